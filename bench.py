#!/usr/bin/env python3
"""bench.py — rendered rays/s of the NeRF-H hot path on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch: the test-time render of ONE 640x480 frame
(307,200 rays, 64 coarse + 128 importance samples, NeRF-H D=8 W=128) — BASELINE.json
configs[1], synthetic random-weight scene of SURVEY.md §8(d), inputs (pose, histogram, weights)
resident in HBM before the timed region.  With N > 1 GPUs every rank renders its own block of
K frames of the render_path batch (weak scaling, no data-path collective) and the rendered
frames are gathered to rank 0 over RCCL inside the timed region.

`python bench.py --gpus N` launches its own N ranks (torch.distributed.run, 127.0.0.1) when it is not
already running under a launcher (WORLD_SIZE unset); under `python -m torch.distributed.run ... bench.py
--gpus N` it reads RANK / LOCAL_RANK / WORLD_SIZE as given.

Prints ONE JSON line (rank 0):
  value / ms_per_step   the headline, at REFERENCE precision (the reference computes in fp32): `--precision f16x3` (default) =
                        split-f16, fp32-grade — every operand hi + lo in f16, three f16 MFMAs per product, fp32 accumulate —
                        admitted IN THE SAME RUN by `cpu_baseline.parity_vs_oracle` and by `config.fp32_grade_check` (the fine
                        network's 9 raw channels, split-f16 vs the exact-fp32 MFMA kernel on the same samples);
  precisions            the same workload on the other arithmetic modes — "f32" (exact fp32 MFMA), "f16" (f16 MFMA inputs:
                        narrower than the reference, the fast option) and split-f16 fine + f16 coarse — each with value,
                        ms_per_step, the fine kernel's roofline against ITS peak, and parity vs the oracle;
  roofline              dominant kernel (fine MLP, MFMA bound): algorithmic FLOPs per launch / average launch duration
                        measured with HIP events on the launch stream (dfn_profile_*);
  hbm                   achieved GB/s of the HBM-bound stage kernels (sampling, ray bias, compositing), same events;
  secondary             BASELINE configs[3] (DFNet forward ms / 480x640 image), configs[4] (DFNet_dm step ms at the per-GPU
                        shape), DFNet's own training step (N2), the NeRF-H optimisation step and a netwidth-256 frame, each with its parity
                        number against the oracle;
  cpu_baseline          the oracle (torch-CPU port of the reference path) on this box's host cores as SURVEY §8(d) defines it:
                        median of 3 runs over a 16,384-ray subset of the judged frame at the fastest thread count of a sweep,
                        plus one full 160x120 frame at 32+64 samples (BASELINE configs[0] shape) (rank 0, N = 1 only).
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, W, FOCAL, NEAR, FAR = 480, 640, 585.0, 0.0, 2.5
NC, NI = 64, 128
MAC_COARSE, MAC_FINE = 130944, 182720  # algorithmic MAC per sample, SURVEY.md Appendix A
PEAK_TFLOPS = {"f16": 2500.0, "f32": 157.3, "f16x3": 2500.0 / 3}  # dense MFMA peaks, MI355X_MICROARCH.md (split-f16: 3 f16 MFMAs per product)
HBM_PEAK_GBS = 8000.0
PREC_TEXT = {"f16": "f16 MFMA inputs / fp32 accumulate", "f32": "exact fp32 MFMA",
             "f16x3": "split-f16: hi/lo f16 operands, 3 f16 MFMAs per product, fp32 accumulate (fp32-grade)"}
PREC_GATE = {
    "f16x3": "split-f16, fp32-grade: the reference computes in fp32; every product here is hi*hi + hi*lo + lo*hi of f16 halves "
             "accumulated in fp32 (the dropped lo*lo term is 2^-22 relative), checked in this run against the exact-fp32 MFMA kernel "
             "(config.fp32_grade_check) and against the fp32 oracle (cpu_baseline.parity_vs_oracle)",
    "f32": "exact fp32 MFMA (v_mfma_f32_32x32x2_f32): the reference's own arithmetic",
    "f16": "NARROWER than the reference (f16 MFMA inputs, fp32 accumulate): admitted only by the in-run parity check against the "
           "fp32 oracle (north_star tolerance 1e-3); not a reference-precision number"}
# profile slots of dfn_profile_read (include/dfnet_hip.h: DFN_PROF_*)
P_COARSE, P_FINE, P_SAMPLE, P_RAYBIAS, P_COMBINE, P_COMPOSITE = range(6)


PREC_KERNEL_TAG = {"f16": "PrecF16", "f32": "PrecF32", "f16x3": "PrecX3"}


# What bounds sample_fine_kernel, from its counters (round-6 correction of "HBM traffic = the algorithmic bytes", which the committed
# PMC record contradicted).  Filled in by the traffic record: see hbm_records().
SAMPLE_FINE_NOTE = ("VALU-issue-bound wave-per-ray scans (~440 vector instructions per ray: inverse-CDF search, scans, rank merge), about two "
                    "thirds of the SIMD issue time.  Its `traffic` is NOT the algorithmic 1 024 B/ray: the counters show ~3.6x of it "
                    "(WRITE_SIZE alone 2.5x the 768 B/ray of z_fine) — the kernel keeps its per-lane merge state in scratch (private memory), "
                    "which spills through L2 to HBM.  0.35 ms of a 61 ms frame; see DESIGN.md section 3.2 for the no-scratch variant's numbers")


def pmc_traffic(kernel="nerfh_fine_kernel", precision=None):
    """HBM bytes per launch of a kernel from the committed rocprofv3 PMC summaries (profiles/*_pmc_summary*.json: separate
    --pmc passes of this same command per precision, tools/gpu_round.sh).  FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950
    FETCH_SIZE reports half of a wide coalesced read stream (MI355X_MICROARCH.md §HBM), so it is doubled.
    `precision` selects the template instantiation (PrecF16 / PrecX3 / PrecF32 in the kernel name).  None when no summary
    holds the kernel."""
    import glob
    tag = PREC_KERNEL_TAG.get(precision)
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary*.json")), reverse=True):
        for name, c in json.load(open(f)).items():
            if kernel in name and (tag is None or tag in name) and "FETCH_SIZE" in c and "WRITE_SIZE" in c and c.get("dispatches"):
                return (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 / c["dispatches"]
    return None


# ---------------------------------------------------------------------------------------------- CPU baseline (oracle)
def oracle_sample(sample_rays):
    """(rows, weights) of `sample_rays` random rays of frame 0 for the oracle."""
    from dfnet_amd import synthetic as syn
    from oracle import nerfh_oracle as orc
    T = torch.from_numpy
    cw, fw, ea, et = syn.nerfh_weights(0)
    c = {k: T(v) for k, v in cw.items()}
    f = {k: T(v) for k, v in fw.items()}
    ro, rd = orc.get_rays(H, W, FOCAL, T(syn.orbit_pose(0, 8))[:3, :4])
    sel = torch.randperm(H * W, generator=torch.Generator().manual_seed(0))[:sample_rays]
    rows = orc.pack_ray_rows(ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel], NEAR, FAR, syn.HIST_IDX)
    return rows, (c, f, T(ea), T(et))


def numa0_cpus():
    """CPU ids of NUMA node 0 (the baseline's threads are pinned there: one memory domain, no migration); all CPUs if unknown."""
    try:
        txt = open("/sys/devices/system/node/node0/cpulist").read().strip()
        cpus = set()
        for part in txt.split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        allowed = os.sched_getaffinity(0)
        cpus &= allowed
        return sorted(cpus) if cpus else sorted(allowed)
    except OSError:
        return sorted(os.sched_getaffinity(0))


def physical_cores(cpus):
    """One CPU id per physical core among `cpus` (SMT siblings dropped), in id order."""
    seen, out = set(), []
    for c in sorted(cpus):
        try:
            sib = open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read().strip()
        except OSError:
            sib = str(c)
        if sib not in seen:
            seen.add(sib)
            out.append(c)
    return out


def cpu_baseline_worker(mode, sample_rays, out_path):
    """Runs in its own process (cpu_baseline() below; the parent sets the affinity before exec, OMP_PROC_BIND=close / OMP_PLACES=cores).
    mode "sweep": the oracle's rays/s on an eighth of the sample per thread count (the oracle is one torch-CPU process: 128 threads are
    several times SLOWER than 8-16 here) -> {threads: rays/s}.  mode "run": this process owns exactly as many physical cores as
    threads (none of them among the first eight of the node, where the OS and the parent live): ONE untimed full pass over the
    `sample_rays`-ray subset of frame 0 at the judged 64+128 samples (allocator, thread pool, caches: without it three timed runs
    fell monotonically 11.8 -> 7.2 s), then five timed passes — `value` = the median, the minimum beside it — plus one full 160x120
    frame at 32+64 samples (BASELINE configs[0] shape).  Writes its record (and rows / reference outputs) to out_path."""
    from dfnet_amd import synthetic as syn
    from oracle import nerfh_oracle as orc
    ncpu = int(os.environ.get("DFN_CPU_PIN_COUNT", "0")) or len(os.sched_getaffinity(0))
    rows, (c, f, ea, et) = oracle_sample(sample_rays)
    if mode == "sweep":
        # passes of the SAME size as the timed ones (a quarter of the sample): the oracle's rays/s depends on the chunk (1 024-ray passes
        # ranked 8 threads above 16, 4 096-ray passes the other way round by 1.5 x)
        cand = sorted({t for t in (8, 16, 32) if 1 <= t <= ncpu} | {min(ncpu, 8)})
        q = max(64, sample_rays // 4)
        sub = rows[: 2 * q]
        sweep = {}
        with torch.no_grad():
            for t in cand:
                torch.set_num_threads(t)
                orc.render_rays(sub[:256], c, f, ea, et, NC, NI)  # warm-up (thread pool, allocator)
                best = float("inf")
                for k in range(2):                                # the faster of two passes: one pass per setting let the shared host pick the count
                    part = sub[k * q:(k + 1) * q]
                    t0 = time.perf_counter()
                    orc.render_rays(part, c, f, ea, et, NC, NI)
                    best = min(best, (time.perf_counter() - t0) / part.shape[0])
                sweep[t] = 1.0 / best
        torch.save({"sweep": sweep, "sub": sub.shape[0]}, out_path)
        return
    threads = ncpu
    torch.set_num_threads(threads)
    # The host is shared (other tenants' jobs come and go on the same sockets): five 4-7 s passes over the whole sample spread 2.7 ... 7.1 s
    # on one box.  Interference only ever slows a pass down, so the sample is timed as twenty-four short passes (a quarter of the rows each,
    # the four quarters in turn, ~1-2 s) and `value` is the median of the FASTEST FIVE — the estimator `timeit` uses, stated here; every
    # pass time is in the record, and so is the plain median over all of them.
    q = sample_rays // 4
    N_PASS = 24
    with torch.no_grad():
        t0 = time.perf_counter()
        parts = [orc.render_rays(rows[k * q:(k + 1) * q], c, f, ea, et, NC, NI) for k in range(4)]   # untimed: warm-up + the reference outputs
        warm = time.perf_counter() - t0
        ref = {k: torch.cat([p_[k] for p_ in parts]) for k in parts[0]}
        rows = rows[:4 * q]
        runs = []
        for i in range(N_PASS):
            k = i % 4
            t0 = time.perf_counter()
            orc.render_rays(rows[k * q:(k + 1) * q], c, f, ea, et, NC, NI)
            runs.append(time.perf_counter() - t0)
        best5 = sorted(runs)[:5]
        dt = best5[2]
        dt_all = sorted(runs)[N_PASS // 2]
        # one full frame of BASELINE configs[0]'s shape (160x120, 32+64 samples), fastest of three passes after a warm-up
        pose0 = torch.from_numpy(syn.orbit_pose(0, 8))
        orc.render(120, 160, FOCAL / 4, 32768, c, f, ea, et, 32, 64, NEAR, FAR, syn.HIST_IDX, c2w=pose0)
        fr = []
        for _ in range(3):
            t0 = time.perf_counter()
            orc.render(120, 160, FOCAL / 4, 32768, c, f, ea, et, 32, 64, NEAR, FAR, syn.HIST_IDX, c2w=pose0)
            fr.append(time.perf_counter() - t0)
        dt_frame = min(fr)
    rec = {"value": q / dt, "min_time_value": q / min(runs), "median_of_all_passes_value": q / dt_all, "unit": "rays/s", "cores": threads,
           "kind": "port",
           "sample": f"{N_PASS} passes of {q} rays each (the four quarters of {sample_rays} random rays of frame 0 in turn, 64+128 samples, one "
                     f"chunk per pass) after one untimed pass over all of them ({warm:.1f} s); value = the median of the FASTEST FIVE passes "
                     f"({', '.join('%.2f' % r for r in best5)} s: spread {(best5[4] - best5[0]) / dt * 100:.1f} % of their median) — the host is "
                     f"shared and interference only slows a pass; all {N_PASS}: {', '.join('%.2f' % r for r in runs)} s "
                     f"(oracle/nerfh_oracle.py, torch {torch.__version__} CPU fp32, autograd anomaly mode off, no_grad)",
           "pinning": f"own process, affinity = {threads} physical cores of NUMA node 0 (set before exec; not the node's first eight), "
                      f"OMP_PROC_BIND={os.environ.get('OMP_PROC_BIND')} OMP_PLACES={os.environ.get('OMP_PLACES')}",
           "runs_s": [round(r, 3) for r in runs],
           "fastest_five_spread": (best5[4] - best5[0]) / dt,
           "frame_640x480_extrapolated_s": H * W / (q / dt),
           "full_frame_160x120_32+64": {"seconds": dt_frame, "rays_per_s": 160 * 120 / dt_frame, "chunk": 32768, "passes_s": [round(r, 3) for r in fr]},
           "host_cpus": os.cpu_count()}
    torch.save({"rec": rec, "rows": rows, "ref": ref}, out_path)


def cpu_baseline(sample_rays):
    """SURVEY §8(d) CPU leg in child processes pinned inside one NUMA node (cpu_baseline_worker): a thread-count sweep, then the timed
    runs in a process that owns exactly the fastest count of physical cores.  Returns (record, (rows, reference outputs on the subset))."""
    import tempfile
    out = os.path.join(tempfile.mkdtemp(prefix="dfn_cpu_"), "cpu.pt")
    cores = physical_cores(numa0_cpus())
    pool = cores[8:] if len(cores) >= 16 else cores     # leave the node's first cores to the OS, IRQs and this process

    def child(mode, cpus):
        env = dict(os.environ, OMP_PROC_BIND="close", OMP_PLACES="cores", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="",
                   DFN_CPU_PIN_COUNT=str(len(cpus)))
        env.pop("OMP_NUM_THREADS", None)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-worker", mode, str(sample_rays), out], env=env, capture_output=True,
                           text=True, preexec_fn=lambda: os.sched_setaffinity(0, cpus))   # set before exec: the OpenMP runtime binds inside it
        if r.returncode != 0:
            raise RuntimeError(f"cpu baseline worker ({mode}) failed:\n" + r.stderr[-2000:])
        d = torch.load(out, weights_only=False)
        os.remove(out)
        return d

    sw = child("sweep", pool)
    best = max(sw["sweep"], key=sw["sweep"].get)
    d = child("run", pool[:best])
    rec = d["rec"]
    rec.update({"numa0_physical_cores": len(cores), "thread_sweep_rays_per_s": {str(k): round(v, 1) for k, v in sw["sweep"].items()},
                "sweep_sample": f"the faster of two passes of {sw['sub'] // 2} rays (the timed passes' size) per setting in a process pinned to "
                                f"{len(pool)} cores; `cores` = the fastest setting, used for `value`"})
    return rec, (d["rows"], d["ref"])


def fp32_grade_check(E, dev, n=4096):
    """What admits split-f16 as reference-precision arithmetic: the fine network's raw output (9 channels per sample) on the SAME
    samples from the split-f16 kernel and from the exact-fp32 MFMA kernel (v_mfma_f32_32x32x2_f32), n rays x 192 samples."""
    from dfnet_amd import engine as eng, synthetic as syn
    o, d, v = eng.raygen(H, W, FOCAL, torch.from_numpy(syn.orbit_pose(0, 8)).to(dev))
    sel = torch.randperm(H * W, generator=torch.Generator().manual_seed(1))[:n].to(dev)
    o, d, v = (t.reshape(-1, 3)[sel].contiguous() for t in (o, d, v))
    hist = torch.from_numpy(syn.HIST_IDX).to(dev)
    sigma = E.mlp_coarse(o, d, NC, NEAR, FAR, precision="f32")
    z = eng.sample_fine(sigma, NI, NEAR, FAR)
    r32 = E.mlp_fine(o, d, v, hist, z, precision="f32").double()
    r3 = E.mlp_fine(o, d, v, hist, z, precision="f16x3").double()
    diff = (r3 - r32).abs().reshape(-1, 9)
    scale = r32.abs().reshape(-1, 9).amax(0).clamp_min(1e-30)
    return {"raw_max_rel_f16x3_vs_f32": float((diff.amax(0) / scale).max()),
            "raw_rms_rel_f16x3_vs_f32": float((diff.pow(2).mean(0).sqrt() / scale).max()),
            "per": "channel: max |f16x3 - f32| / max |f32| over the channel, worst of the 9 raw channels",
            "rays": n, "samples_per_ray": NC + NI}


def parity(engine, rows, ref, precision, dev):
    from dfnet_amd import synthetic as syn
    rgb, disp, acc, _ = engine.render_rays(rows[:, 0:3].to(dev), rows[:, 3:6].to(dev), torch.from_numpy(syn.HIST_IDX).to(dev),
                                           NC, NI, NEAR, FAR, precision=precision)
    d = (rgb.cpu() - ref["rgb_map"]).double()
    return {"psnr_db": float(-10 * torch.log10((d ** 2).mean().clamp_min(1e-30))),
            "rgb_max_rel": float(d.abs().max() / ref["rgb_map"].abs().max()),
            "disp_max_rel": float((disp.cpu() - ref["disp_map"]).abs().max() / ref["disp_map"].abs().max()),
            "acc_max_rel": float((acc.cpu() - ref["acc_map"]).abs().max() / ref["acc_map"].abs().max()),
            "rays": int(rows.shape[0]), "tolerance": 1e-3}


# ---------------------------------------------------------------------------------------------- timed render
def read_profile(lib):
    from dfnet_amd import _lib
    out = {}
    for slot in range(6):
        ms, n = ctypes.c_double(), ctypes.c_int()
        _lib.check(lib.dfn_profile_read(slot, ctypes.byref(ms), ctypes.byref(n)), "dfn_profile_read")
        out[slot] = (ms.value, n.value)
    return out


def load_probe():
    """tools/probe/libdfn_probe.so: the bench-only helper library with the bare MFMA loop (built by `make -C dfnet_amd/csrc`, NOT part
    of libdfnet_hip.so)."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "probe", "libdfn_probe.so")
    lib = ctypes.CDLL(path)
    lib.dfn_probe_mfma_rate.restype = ctypes.c_int
    lib.dfn_probe_mfma_rate.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.POINTER(ctypes.c_double), ctypes.c_void_p]
    lib.dfn_probe_last_error.restype = ctypes.c_char_p
    return lib


def sustained_mfma(lib, stream=None, seconds=0.4):
    """What the matrix pipe of THIS box sustains (dfn_probe_mfma_rate: nothing but independent dense-f16 MFMAs on every SIMD): with
    zero operands the nominal peak, with random operands the rate once power management has settled the clock.  The MLP kernels'
    MFMA work (x3 for split-f16: three f16 MFMAs per product) is priced against the random-operand figure as `frac_of_sustained`."""
    probe = load_probe()
    out = {}
    for key, rnd in (("zero_operands_TFLOPs", 0), ("random_operands_TFLOPs", 1)):
        tf = ctypes.c_double()
        rc = probe.dfn_probe_mfma_rate(rnd, float(seconds), ctypes.byref(tf), None)
        if rc != 0:
            raise RuntimeError(f"dfn_probe_mfma_rate failed ({rc}): {probe.dfn_probe_last_error().decode()}")
        out[key] = tf.value
    out["note"] = ("dense f16 v_mfma_f32_32x32x16_f16 back to back on all SIMDs for %.1f s each, measured in this run; roofline.peak is the "
                   "nominal 2.4 GHz figure, which this part reaches only with operands that do not toggle" % seconds)
    return out


SUSTAINED = {}   # filled by main() from sustained_mfma(): the secondary records price their f16 MFMA work against it too


def of_sustained(f16_mfma_tflops):
    """f16-MFMA TFLOP/s of a workload / the random-operand MFMA rate measured in this run (None before the probe ran)."""
    r = SUSTAINED.get("random_operands_TFLOPs")
    return f16_mfma_tflops / r if r else None


def add_sustained(roof, sus, precision):
    """frac_of_sustained: the kernel's f16-MFMA work rate / the random-operand MFMA rate of this box (f32 kernels: not power-limited, skipped)."""
    if precision in ("f16", "f16x3") and sus.get("random_operands_TFLOPs"):
        mult = 3.0 if precision == "f16x3" else 1.0
        roof["frac_of_sustained_mfma"] = roof["achieved"] * mult / sus["random_operands_TFLOPs"]


class GpuSampler:
    """Package power (W) and shader clock (MHz) of ONE device, sampled on a thread while the timed region runs (what explains a
    bent scaling curve: a node of eight MI355X at their 1.4 kW caps).  sysfs hwmon of the device's PCI function when it is there
    (power1_average in microwatts, freq1_input in Hz), `rocm-smi -d i --json` otherwise; every field None when neither answers."""

    def __init__(self, index, period=0.05):
        import glob
        import threading
        self.index, self.period = index, period
        self.power, self.sclk = [], []
        self._stop = threading.Event()
        self._hw = None
        try:
            p = torch.cuda.get_device_properties(index)
            bdf = f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
            hw = glob.glob(f"/sys/bus/pci/devices/{bdf}/hwmon/hwmon*")
            if hw and (os.path.exists(hw[0] + "/power1_average") or os.path.exists(hw[0] + "/power1_input")):
                self._hw = hw[0]
        except Exception:   # noqa: BLE001 - the sampler is telemetry: never the reason a bench fails
            self._hw = None
        self._thread = threading.Thread(target=self._run, daemon=True)

    def _read_sysfs(self):
        def num(name):
            try:
                with open(f"{self._hw}/{name}") as f:
                    return float(f.read().strip())
            except OSError:
                return None
        pw = num("power1_average")
        if pw is None:
            pw = num("power1_input")
        fq = num("freq1_input")
        return (None if pw is None else pw / 1e6), (None if fq is None else fq / 1e6)

    def _read_smi(self):
        try:
            out = subprocess.run(["rocm-smi", "-d", str(self.index), "--showpower", "--showclocks", "--json"], capture_output=True,
                                 text=True, timeout=5).stdout
            card = next(iter(json.loads(out).values()))
            pw = next((float(v) for k, v in card.items() if "power" in k.lower() and "max" not in k.lower()), None)
            ck = next((v for k, v in card.items() if "sclk" in k.lower()), None)
            fq = float("".join(ch for ch in str(ck).split("Mhz")[0] if ch.isdigit() or ch == ".")) if ck else None
            return pw, fq
        except Exception:   # noqa: BLE001
            return None, None

    def _run(self):
        while not self._stop.is_set():
            pw, fq = self._read_sysfs() if self._hw else self._read_smi()
            if pw is not None:
                self.power.append(pw)
            if fq is not None:
                self.sclk.append(fq)
            self._stop.wait(self.period if self._hw else 0.5)

    def __enter__(self):
        self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._thread.join(timeout=10)

    def means(self):
        m = lambda v: (sum(v) / len(v)) if v else float("nan")
        return m(self.power), m(self.sclk), float(len(self.power))


def per_rank_record(stats, gathered_bytes):
    """The N-rank run explained: per-rank render / gather seconds and mean clock / power over the timed region."""
    cols = ("render_s", "gather_s", "power_w", "sclk_mhz", "samples")
    rec = {}
    for i, c in enumerate(cols[:4]):
        v = [float(x) for x in stats[:, i]]
        ok = [x for x in v if x == x]
        rec[c] = {"per_rank": [None if x != x else round(x, 5) for x in v], "min": min(ok) if ok else None, "max": max(ok) if ok else None,
                  "mean": sum(ok) / len(ok) if ok else None}
    rec["samples_per_rank"] = [int(x) for x in stats[:, 4]]
    rec["gathered_bytes"] = int(gathered_bytes)
    return rec


def timed_render(E, lib, precision, poses, hist, rgbs, disps, acc, K, Wm, world=1, gather=None):
    """Wm untimed + K timed frames; returns (seconds for the K frames, max over ranks; per-kernel HIP-event averages; [world, 5]
    per-rank record: this rank's render seconds, gather seconds, mean package power, mean shader clock, sample count)."""
    from dfnet_amd import dist as ddist
    nf = poses.shape[0]

    def step(k):
        E.render_image(poses[k % nf], H, W, FOCAL, hist, NC, NI, NEAR, FAR, precision=precision,
                       out=(rgbs[k % nf], disps[k % nf], acc))

    for k in range(Wm):
        step(k)
    if ddist.active() and gather is not None:  # warm the collective too: communicator set-up (the root's receive buffers exist already)
        gather()
    torch.cuda.synchronize()
    ddist.barrier()
    lib.dfn_profile_enable(1)
    torch.cuda.synchronize()
    with GpuSampler(rgbs.device.index or 0) as smp:
        t0 = time.perf_counter()
        for k in range(K):
            step(k)
        torch.cuda.synchronize()
        t_render = time.perf_counter() - t0
        if ddist.active() and gather is not None:
            gather()
            torch.cuda.synchronize()
        t_gather = time.perf_counter() - t0 - t_render
        ddist.barrier()
        dt = time.perf_counter() - t0
    dt = ddist.max_over_ranks(dt, rgbs.device)
    prof = read_profile(lib)
    lib.dfn_profile_enable(0)
    pw, ck, ns = smp.means()
    stats = ddist.all_gather_floats([t_render, t_gather, pw, ck, ns], rgbs.device)
    return dt, prof, stats


def mlp_roofline(prof, K, precision, value_per_gpu):
    rays = H * W
    ms, n = prof[P_FINE]
    flops = 2.0 * MAC_FINE * (NC + NI) * rays * K / max(n, 1)
    achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    peak = PEAK_TFLOPS[precision]
    rec = {"bound": "mfma", "kernel": "nerfh_fine_kernel", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
           "frac": achieved / peak, "launches_per_step": n / K, "avg_launch_ms": ms,
           "algorithmic_flops_per_launch": flops, "coarse_kernel_avg_launch_ms": prof[P_COARSE][0],
           "whole_path_mfma_frac": value_per_gpu * 2.0 * (MAC_COARSE * NC + MAC_FINE * (NC + NI)) / 1e12 / peak}
    if precision == "f16x3":
        # two honest readings of the same launch time.  `frac` = MFMA-ISSUE utilisation: split-f16 issues three f16 MFMAs per fp32-grade
        # product, so its peak is 2 500 / 3 algorithmic TFLOP/s.  SURVEY 8(d)'s formula prices the ALGORITHMIC FLOPs against the dense f16
        # peak of the instruction class (2 500): a third of `frac`.
        rec["frac_is"] = "issued: algorithmic TFLOP/s x 3 f16 MFMAs per product / 2 500 (= achieved / peak with peak = 2 500 / 3)"
        rec["frac_algorithmic_of_f16_peak"] = achieved / PEAK_TFLOPS["f16"]
    return rec


def hbm_records(prof, K, E, dev):
    """Achieved HBM GB/s of the bandwidth-bound stage kernels: algorithmic bytes per launch (SURVEY.md §8(d) per-ray bytes x
    rays per launch) / average launch duration (HIP events on the launch stream)."""
    from dfnet_amd import engine as eng
    rays = H * W
    out = {}
    for name, slot, bpr, what in (
            ("sample_fine_kernel", P_SAMPLE, 256 + 768, "256 B sigma in + 768 B z_fine out per ray"),
            ("ray_bias_kernel", P_RAYBIAS, 12 + 512, "12 B viewdir in + 512 B bias table out per ray"),
            ("composite_combine_kernel", P_COMBINE, 144 + 20, "144 B segment composites in + 20 B rgb/disp/acc out per ray")):
        ms, n = prof[slot]
        if n:
            b = bpr * rays * K / n
            out[name] = {"avg_launch_ms": ms, "algorithmic_bytes_per_launch": b, "achieved_GBps": b / (ms * 1e-3) / 1e9,
                         "frac_of_hbm_peak": b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "bytes_per_ray": what,
                         "traffic": pmc_traffic(name),   # HBM bytes per launch from the committed PMC passes (as roofline.traffic)
                         "launches_per_step": n / K}
    # what the PMC passes say about the two that sit near 10 % of the HBM rate: their traffic is the algorithmic bytes, they are bound
    # by instruction issue, not by memory (profiles/*_pmc_summary.json: SQ_INSTS_VALU per launch x 4 cycles over the SIMD time)
    notes = {"sample_fine_kernel": SAMPLE_FINE_NOTE,
             "ray_bias_kernel": "VALU-bound: 6 208 fp32 MACs per ray for the two per-ray tables (12.4 KFLOP), HBM traffic 1.07x the algorithmic bytes"}
    for k, v in notes.items():
        if k in out:
            out[k]["limited_by"] = v
    # the raw-path compositor (runs when the caller asks for `raw` / on the gradient path): one 61,440-ray pass of random raw
    n, Nf = 61440, NC + NI
    g = torch.Generator(device=dev).manual_seed(0)
    raw = torch.rand(n, Nf, 9, device=dev, generator=g)
    z = torch.sort(torch.rand(n, Nf, device=dev, generator=g) * 2.5, dim=-1)[0]
    eng.composite_fine(raw, z)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    reps = 10
    e0.record()
    for _ in range(reps):
        eng.composite_fine(raw, z)   # allocates 3 small outputs per call (torch caching allocator, no sync)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    b = (6912 + 768 + 20) * n
    out["composite_fine_kernel"] = {"avg_launch_ms": ms, "algorithmic_bytes_per_launch": b, "achieved_GBps": b / (ms * 1e-3) / 1e9,
                                    "frac_of_hbm_peak": b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                    "bytes_per_ray": "6912 B raw + 768 B z in, 20 B out per ray (raw path only; the default "
                                                     "render composites inside the fine kernel)",
                                    "note": "torch events around 10 back-to-back launches on a 61,440-ray pass"}
    out["peak_GBps"] = HBM_PEAK_GBS
    out["note"] = ("sample_fine / ray_bias are one-wave-per-ray kernels whose time goes to serial wave scans, binary searches and the "
                   "sort in LDS (PMC: SQ_WAIT_ANY / SQ_WAVE_CYCLES = 58 %), not to HBM: their GB/s is reported because north_star asks for "
                   "it; together they are 3 % of a frame")
    return out


# ---------------------------------------------------------------------------------------------- secondary workloads
def secondary_dfnet(dev):
    """BASELINE configs[3] as SURVEY 8(d) C4 defines it: 256 frames of 480x640 RENDERED BY THE NeRF PATH (the judged synthetic
    scene, orbit of 256 poses) streamed through DFNet.forward(return_feature=True, isSingleStream=True, return_pose=False,
    upsample 480x640) in batches of featurenet_batch_size = 4; parity = relative L2 per pyramid level against the CPU oracle,
    accumulated on the device over a 16-frame subset (the output is 472 MB per frame)."""
    from dfnet_amd import engine as eng, synthetic as syn
    from oracle import dfnet_oracle as dor
    w = syn.dfnet_weights(3)
    E = eng.DfnetEngine(3, 12).load_numpy(w)
    B, NF = 4, 256
    # the frames: NeRF-H renders (f16 arithmetic: they are inputs here), [NF, 3, 480, 640] in [0, 1]
    cw, fw, ea, et = syn.nerfh_weights(0)
    R = eng.NerfHEngine(precision="f16").load_numpy(cw, fw, ea, et)
    hist = torch.from_numpy(syn.HIST_IDX).to(dev)
    frames = torch.empty(NF, 3, H, W, device=dev)
    t0 = time.perf_counter()
    for k in range(NF):
        rgb, _, _ = R.render_image(torch.from_numpy(syn.orbit_pose(k, NF)).to(dev), H, W, FOCAL, hist, NC, NI, NEAR, FAR)
        frames[k].copy_(rgb.permute(2, 0, 1).clamp(0, 1))
    torch.cuda.synchronize()
    render_s = time.perf_counter() - t0
    del R
    x = frames[:B].contiguous()
    out = {"workload": f"BASELINE configs[3] / SURVEY 8(d) C4: DFNet.forward(return_feature=True, isSingleStream=True, return_pose=False, "
                       f"upsample 480x640) on {NF} NeRF-rendered 480x640 frames in batches of {B}; 325.3 GFLOP algorithmic per image",
           "frames_rendered_in_s": render_s, "precisions": {}}
    for prec, reps in (("f16x3", 10), ("f16", 5), ("f32", 2)):
        for _ in range(2):
            E.forward(x, True, True, False, 480, 640, precision=prec)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            E.forward(x, True, True, False, 480, 640, precision=prec)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps / B
        mf = {"f16": 1, "f16x3": 3, "f32": 1}[prec]
        out["precisions"][prec] = {"ms_per_image": dt * 1e3, "algorithmic_TFLOPs": 325.3e9 / dt / 1e12,
                                   "mfma_frac": 325.3e9 * mf / dt / 1e12 / (157.3 if prec == "f32" else 2500.0),
                                   "arithmetic": PREC_TEXT[prec]}
        if prec != "f32":
            out["precisions"][prec]["mfma_frac_of_sustained"] = of_sustained(325.3e9 * mf / dt / 1e12)
    # the whole stream, default arithmetic
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(0, NF, B):
        E.forward(frames[k:k + B], True, True, False, 480, 640, precision="f16x3")
    torch.cuda.synchronize()
    dts = time.perf_counter() - t0
    out["stream_256_frames_f16x3"] = {"seconds": dts, "frames_per_s": NF / dts, "ms_per_image": dts / NF * 1e3,
                                      "mfma_frac": 325.3e9 * 3 * NF / dts / 1e12 / 2500.0,
                                      "mfma_frac_of_sustained": of_sustained(325.3e9 * 3 * NF / dts / 1e12)}
    # the reference's config names "featurenet_batch_size=4 # batch size, 4 or 8" (config_dfnet.txt:17): the other one
    t0 = time.perf_counter()
    for k in range(0, 64, 8):
        E.forward(frames[k:k + 8], True, True, False, 480, 640, precision="f16x3")
    torch.cuda.synchronize()
    dt8 = (time.perf_counter() - t0) / 64
    out["batch_8_f16x3"] = {"ms_per_image": dt8 * 1e3, "mfma_frac": 325.3e9 * 3 / dt8 / 1e12 / 2500.0}
    # parity: 16 of the 256 frames against the CPU oracle; ||F - F_ref||^2 and ||F_ref||^2 per level accumulated on the device in fp64
    sub = list(range(0, NF, NF // 16))
    wt = {k: torch.from_numpy(v) for k, v in w.items()}
    num = {p: torch.zeros(3, dtype=torch.float64, device=dev) for p in ("f16x3", "f16", "f32")}
    den = torch.zeros(3, dtype=torch.float64, device=dev)
    cpu_s = 0.0
    for k in sub:
        with torch.no_grad():
            t0 = time.perf_counter()
            ref = dor.dfnet_forward(wt, frames[k:k + 1].cpu(), True, True, False, 480, 640)[0][0]
            cpu_s += time.perf_counter() - t0
        ref = ref.to(dev)
        den += ref.double().pow(2).sum(dim=tuple(range(1, ref.dim())))
        for prec in num:
            got = E.forward(frames[k:k + 1], True, True, False, 480, 640, precision=prec)[0]
            num[prec] += (got - ref).double().pow(2).sum(dim=tuple(range(1, ref.dim())))
        del ref, got
    for prec in num:
        out["precisions"][prec]["rel_l2_per_level_vs_oracle"] = [float(v) for v in (num[prec] / den).sqrt()]
    out["parity_frames"] = len(sub)
    out["default_precision"] = "f16x3"
    out["ms_per_image"] = out["stream_256_frames_f16x3"]["ms_per_image"]
    out["cpu_oracle_s_per_image"] = cpu_s / len(sub)
    return out


def secondary_dm_step(dev):
    """BASELINE configs[4] at its per-GPU shape: DFNet_dm step, batch 4, 240x320, render 60x80 at 64+128 + bicubic x4,
    level-0 feature loss; parity = the step's loss against the composition of the CPU oracles on the same inputs."""
    from types import SimpleNamespace
    from dfnet_amd import engine as eng, optim, synthetic as syn
    from dfnet_amd.dfnet import DFNet
    import dfnet_amd.direct_feature_matching as dfm
    from dfnet_amd.direct_feature_matching import matching_step_grad, train_on_batch, train_on_batch_device
    from dfnet_amd.nerfw import HipQuery
    from oracle import dfnet_oracle as dor, nerfh_oracle as orc
    T = torch.from_numpy
    B, Hh, Ww, focal = 4, 240, 320, 585.0 / 2
    w = syn.dfnet_weights(3)
    sd = {k: T(v) for k, v in w.items()}
    model, feat_model = DFNet().to(dev).eval(), DFNet().to(dev).eval()
    model.load_state_dict({k: v.to(dev) for k, v in sd.items()}, strict=False)
    feat_model.load_state_dict({k: v.to(dev) for k, v in sd.items()}, strict=False)
    for q in feat_model.parameters():
        q.requires_grad_(False)
    cw, fw, ea, et = syn.nerfh_weights(0)
    E = eng.NerfHEngine(precision="f16").load_numpy(cw, fw, ea, et)
    kw = dict(network_query_fn=HipQuery(E), perturb=False, N_importance=NI, N_samples=NC, use_viewdirs=True,
              white_bkgd=False, raw_noise_std=0., test_time=True, ndc=False, lindisp=False, near=NEAR, far=FAR)
    setup = dict(pose_scale=1.0, pose_scale2=1.0, move_all_cam_vec=[0., 0., 1.0])
    args = SimpleNamespace(svd_reg=True, chunk=32768, feature_matching_lvl=[0], per_channel=False, combine_loss=True,
                           combine_loss_w=[0.3, 0.2, 1.0])
    data = torch.rand(B, 3, Hh, Ww, generator=torch.Generator().manual_seed(1)).to(dev)
    gt = torch.stack([T(syn.orbit_pose(k, 8))[:3, :4].reshape(12) for k in range(B)])
    hist = T(syn.HIST_IDX).repeat(B, 1)
    hwf = [Hh, Ww, focal]

    def timed(fn, iters=5):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            o = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3, o

    pose_ms, out = timed(lambda: matching_step_grad(args, data, model, feat_model, gt, hist, hwf, True, dev, setup, **kw))
    opt = optim.Adam(model.parameters(), lr=1e-7)
    # the step as train_on_epoch runs it (losses stay on the device, one wait per epoch); train_on_batch itself returns host floats
    # like the reference and so waits for the device every step: timed separately below
    step = lambda: train_on_batch_device(args, data, model, feat_model, gt, hist, hwf, opt, True, dev, setup, **kw)
    step_host = lambda: train_on_batch(args, data, model, feat_model, gt, hist, hwf, opt, True, dev, setup, **kw)
    # the reference's form first: every pyramid level computed, the loss's level index_selected afterwards; then the shipped default,
    # which computes only the level(s) of feature_matching_lvl (bit-identical loss and gradients: tests/test_gpu_grad.py), alternated
    all_ms, pruned_ms = [], []
    for _ in range(3):
        dfm.PRUNE_FEATURE_LEVELS = False
        all_ms.append(timed(step)[0])
        dfm.PRUNE_FEATURE_LEVELS = True
        pruned_ms.append(timed(step)[0])
    full_ms, full_all_ms = sorted(pruned_ms)[1], sorted(all_ms)[1]
    host_ms = timed(step_host)[0]
    # oracle composition (forward only) from the same predicted pose
    with torch.no_grad():
        t0 = time.perf_counter()
        pose_ = out["pose_pred"].cpu().clone()
        pn = pose_.clone()
        pn[:, :3, 3] *= setup["pose_scale"]
        pn[:, :3, 3] += torch.tensor(setup["move_all_cam_vec"])
        pn[:, :3, 3] *= setup["pose_scale2"]
        c, f = {k: T(x) for k, x in cw.items()}, {k: T(x) for k, x in fw.items()}
        rgbs = []
        for b in range(B):
            r = orc.render(Hh // 4, Ww // 4, focal / 4, 1 << 30, c, f, T(ea), T(et), NC, NI, NEAR, FAR, syn.HIST_IDX, c2w=pn[b])[0]
            rgbs.append(torch.nn.Upsample(size=(Hh, Ww), mode='bicubic')(r.permute(2, 0, 1)[None])[0])
        rgb = torch.stack(rgbs)
        dcpu = data.cpu()
        feats, _ = dor.dfnet_forward(sd, torch.cat([dcpu, rgb]), True, False, False, Hh, Ww)
        ft = feats[0][[0]].permute(1, 0, 2, 3, 4).reshape(B, 128, Hh, Ww)
        fr = feats[1][[0]].permute(1, 0, 2, 3, 4).reshape(B, 128, Hh, Ww)
        fl = torch.stack([1 - torch.nn.functional.cosine_similarity(fr[b].reshape(128, -1), ft[b].reshape(128, -1), dim=1, eps=1e-6).mean()
                          for b in range(B)]).mean()
        ref_loss = float(0.3 * torch.nn.functional.mse_loss(pose_.reshape(B, 12), gt) + 0.2 * ((rgb - dcpu) ** 2).mean() + 1.0 * fl)
        cpu_s = time.perf_counter() - t0
    return {"workload": "BASELINE configs[4] per-GPU shape: DFNet_dm step, batch 4, 240x320 frames, NeRF-H render 60x80 at 64+128 "
                        "+ bicubic x4, level-0 cosine feature loss + photometric + pose terms",
            "forward_backward_to_pose_ms": pose_ms, "full_step_ms": full_ms, "full_step_all_levels_ms": full_all_ms,
            "full_step_returning_host_floats_ms": host_ms,
            "full_step_is": "all 28 regressor gradients + Adam (dfnet_amd.optim.Adam: one dfn_adam_step launch) + device-side re-pack of the updated weights, as train_on_epoch runs "
                            "it: the step's loss / PSNR stay on the device and the host waits once per epoch (train_on_batch_device); "
                            "full_step_returning_host_floats_ms = train_on_batch with the reference's signature (numpy floats: one device wait "
                            "per step).  full_step_ms = the "
                            "shipped default: the frozen feature extractor computes only the pyramid level(s) of feature_matching_lvl = [0] "
                            "(the loss reads no other: the reference computes all three and index_selects, direct_feature_matching.py:354-357); "
                            "full_step_all_levels_ms = all three levels computed as the reference does; same loss and gradients bit for bit; "
                            "medians of three alternations",
            "loss": float(out["loss"]), "oracle_loss": ref_loss,
            "loss_rel_diff_vs_oracle": abs(float(out["loss"]) - ref_loss) / max(abs(ref_loss), 1e-12),
            "cpu_oracle_forward_s": cpu_s,
            "gradient_parity": "tests/test_gpu_grad.py (pose gradient and the 28 parameter gradients vs oracle autograd)"}


def secondary_dfnet_train(dev):
    """one optimisation step of DFNet's own training (run_feature.py:166-230 with config_dfnet.txt: triplet loss
    with in-triplet hard-negative mining, random view synthesis, BatchNorm on batch statistics): siamese forward on [target, render]
    (2B frames), pose forward on B synthesised views, losses, backward of every parameter, Adam, device re-pack.  Parity = the step's
    loss against the CPU oracle's forward in train() mode on the same frames."""
    from dfnet_amd import optim, synthetic as syn
    from dfnet_amd.dfnet import DFNet
    from dfnet_amd.feature_misc import PoseLoss, triplet_loss_hard_negative_mining_plus
    from oracle import dfnet_oracle as dor
    T = torch.from_numpy
    B, Hh, Ww = 4, 240, 320
    w = syn.dfnet_weights(3)
    m = DFNet()
    m.load_state_dict({k: T(v) for k, v in w.items()}, strict=False)
    m.to(dev).train()
    m.pyramid_features = True   # what script/run_feature.py sets under --tripletloss: the triplet loss from the low-resolution pyramid
    opt = optim.Adam(m.parameters(), lr=1e-7)
    g = torch.Generator().manual_seed(1)
    target, rgb, virt = (torch.rand(B, 3, Hh, Ww, generator=g) for _ in range(3))
    pose = torch.stack([T(syn.orbit_pose(k, 8))[:3, :4].reshape(12) for k in range(B)])
    dtarget, drgb, dvirt, dpose = (t.to(dev) for t in (target, rgb, virt, pose))

    def step(update=True):
        if m.pyramid_features:   # script/run_feature.py: the synthesised views' pose forward rides in the siamese encoder pass
            feats, pall = m(torch.cat([dtarget, drgb, dvirt]), True, upsampleH=Hh, upsampleW=Ww, feature_images=2 * B)
            pred, vp = pall[:2 * B], pall[2 * B:]
        else:                    # the reference's two calls (run_feature.py:211, :219)
            feats, pred = m(torch.cat([dtarget, drgb]), True, upsampleH=Hh, upsampleW=Ww)
            _, vp = m(dvirt, False)
        loss = PoseLoss(None, pred, torch.cat([dpose, dpose]), dev) + triplet_loss_hard_negative_mining_plus(feats[1], feats[0], margin=1.0)
        loss = loss + PoseLoss(None, vp, dpose, dev)
        loss.backward()
        if update:
            opt.step()
        opt.zero_grad()
        return loss.detach()

    loss0 = float(step(update=False))   # the untouched weights: the figure the oracle reproduces
    step()
    step()
    torch.cuda.synchronize()
    per_iter = []                       # the reference's form: every step ends in a host read of the loss (loss.item(), run_feature.py:226):
    for _ in range(9):                  # timed one by one, median reported (a single slow iteration doubled the mean of five on one box)
        t0 = time.perf_counter()
        float(step())
        torch.cuda.synchronize()
        per_iter.append((time.perf_counter() - t0) * 1e3)
    ms_host = sorted(per_iter)[len(per_iter) // 2]
    # the shipped epoch loop (script/run_feature.py: _step / _mean_loss): the losses stay on the device, one wait per epoch — three
    # runs of nine steps back to back, median run
    runs = []
    for _ in range(3):
        t0 = time.perf_counter()
        keep = [step() for _ in range(9)]
        torch.cuda.synchronize()
        runs.append((time.perf_counter() - t0) * 1e3 / 9)
    ms = sorted(runs)[1]
    per_iter = runs
    peak_gb = torch.cuda.max_memory_allocated() / 2 ** 30
    # the materialised form (round 5: two [3, B, 128, H, W] stacks, the stack triplet kernels, upsample and its adjoint), same box
    m.pyramid_features = False
    step(); step()
    torch.cuda.synchronize()
    runs_s = []
    for _ in range(3):
        t0 = time.perf_counter()
        keep = [step() for _ in range(9)]
        torch.cuda.synchronize()
        runs_s.append((time.perf_counter() - t0) * 1e3 / 9)
    m.pyramid_features = True
    with torch.no_grad():
        t0 = time.perf_counter()
        p = {k: T(v) for k, v in w.items()}
        feats, pred = dor.dfnet_forward(p, torch.cat([target, rgb]), True, False, True, Hh, Ww, bn_stats=[])
        _, vp = dor.dfnet_forward(p, virt, False, True, True, Hh, Ww, bn_stats=[])
        mse = torch.nn.functional.mse_loss
        ref = float(mse(pred, torch.cat([pose, pose])) + dor.triplet_loss(feats[1], feats[0], margin=1.0, mining=2)[0] + mse(vp, pose))
        cpu_s = time.perf_counter() - t0
    return {"workload": f"one DFNet training step (run_feature.py:166-230): featurenet_batch_size {B} -> {2 * B} siamese + {B} synthesised "
                        f"frames of {Hh}x{Ww}, triplet loss (hard-negative mining, four cases) + pose losses, BatchNorm on batch statistics, "
                        "every parameter gradient, Adam, device re-pack",
            "step_ms": ms, "step_ms_all": [round(x, 2) for x in per_iter], "step_ms_with_a_host_read_of_the_loss_per_step": ms_host,
            "step_ms_is": "the shipped epoch loop (script/run_feature.py): losses stay on the device, the host waits once per epoch — nine steps "
                          "back to back, median of three runs; ..._with_a_host_read...: the reference's loss.item() after every step, median of nine",
            "frames_per_s": 3 * B / ms * 1e3,
            "triplet_loss": "closed form of the low-resolution pyramid (csrc/dfnet_triplet_pyr.hip): no enlarged stacks, no upsample / adjoint; "
                            "the synthesised views' pose forward rides in the siamese encoder pass (one batch of 3 x 4 frames)",
            "step_ms_with_materialised_stacks": sorted(runs_s)[1], "peak_mem_GB": peak_gb,
            "arithmetic": "split-f16 (f16x3) forward, data-gradient and weight-gradient products; fp32 accumulate (fp32-grade)",
            "loss": loss0, "oracle_loss": ref, "loss_rel_diff_vs_oracle": abs(loss0 - ref) / max(abs(ref), 1e-12), "cpu_oracle_forward_s": cpu_s,
            "gradient_parity": "tests/test_gpu_triplet_pyr.py (G16: the reference's DFNet + triplet loss + autograd end to end, 43 gradients), "
                               "tests/test_gpu_dfnet.py / test_gpu_grad.py (G10: 46 gradients of the module's training step; G11: the triplet losses)"}


def secondary_nerfh_train(dev):
    """one NeRF-H optimisation step at the reference's defaults (N_rand 1536 rays, 64+128 samples, netwidth 128,
    perturb 1; run_nerf.py:32-80): forward, fused NerfWLoss, every gradient, Adam.  Parity: a 256-ray step against autograd
    through the CPU oracle (loss terms and the worst relative L2 over the 64 gradient tensors)."""
    from dfnet_amd import engine as eng, nerf_train, optim, synthetic as syn
    from dfnet_amd.nerfw import NeRFW
    from oracle import nerfh_oracle as orc
    T = torch.from_numpy
    cw, fw, ea, et = syn.nerfh_weights(0)
    coarse = NeRFW('coarse', D=8, W=128, skips=[4], in_channels_xyz=63, in_channels_dir=27)
    fine = NeRFW('fine', D=8, W=128, skips=[4], in_channels_xyz=63, in_channels_dir=27, encode_appearance=True, encode_transient=True,
                 in_channels_a=50, in_channels_t=20)
    coarse.load_state_dict({k: T(v) for k, v in cw.items()})
    fine.load_state_dict({k: T(v) for k, v in fw.items()})
    emb_a, emb_t = torch.nn.Embedding(1000, 5), torch.nn.Embedding(1000, 2)
    emb_a.weight.data.copy_(T(ea))
    emb_t.weight.data.copy_(T(et))
    mods = [m.to(dev) for m in (coarse, fine, emb_a, emb_t)]
    E = eng.NerfHEngine(precision="f32").load_numpy(cw, fw, ea, et)
    tr = nerf_train.NerfHTrainer(E, *mods)
    rng = np.random.default_rng(5)
    ro, rd = orc.get_rays(H, W, FOCAL, T(syn.orbit_pose(4, 8))[:3, :4])

    def batch(R):
        sel = rng.choice(H * W, R, replace=False)
        return (ro.reshape(-1, 3)[sel].contiguous(), rd.reshape(-1, 3)[sel].contiguous(),
                T(rng.integers(0, 40, (R, 10)).astype(np.float32)), T(rng.uniform(0, 1, (R, 3)).astype(np.float32)))

    # parity at 256 rays, both implementations of the step (fused register-resident chains = the default at netwidth 128; the
    # layer-by-layer exact-fp32 step)
    R = 256
    o, d, hist, target = batch(R)
    gen = torch.Generator().manual_seed(9)
    draws = (torch.rand(R, NC, generator=gen), torch.randn(R, NC, generator=gen), torch.rand(R, NI, generator=gen))
    rows = torch.cat([o, d, torch.zeros(R, 1), torch.full((R, 1), FAR), d / d.norm(dim=-1, keepdim=True), hist], 1)
    c, f = {k: T(v) for k, v in cw.items()}, {k: T(v) for k, v in fw.items()}
    t0 = time.perf_counter()
    ld_ref, _, g_ref, _ = orc.train_step(rows, target, c, f, T(ea), T(et), NC, NI, *draws, perturb=1., raw_noise_std=1.)
    cpu_s = time.perf_counter() - t0
    par = {}
    for tag, exact, split in (("fused", False, False), ("fused_split", False, True), ("exact", True, False)):
        tr.exact, tr.fused_split = exact, split
        ld, _, _ = tr.train_step(o.to(dev), d.to(dev), hist.to(dev), target.to(dev), NC, NI, NEAR, FAR, perturb=1., raw_noise_std=1.,
                                 draws=tuple(t.to(dev) for t in draws))
        par[tag] = {"worst_rel_l2_over_64_gradients": max(float((p.grad.cpu().double() - g_ref[k].double()).norm() / g_ref[k].double().norm())
                                                           for k, p in zip(tr.names, tr.params)),
                    "worst_loss_term_rel_diff": max(abs(float(ld[k]) - float(ld_ref[k])) / abs(float(ld_ref[k])) for k in ld)}
    tr.flush_range_check()
    range_flags = E.range_flags()
    # timing at the reference's batch
    R = 1536
    o, d, hist, target = (t.to(dev) for t in batch(R))
    opt = optim.Adam(tr.params, lr=5e-4, betas=(0.9, 0.999))

    def step():
        tr.train_step(o, d, hist[:1], target, NC, NI, NEAR, FAR, perturb=1., raw_noise_std=0.)
        opt.step()

    def timed(fn, n=10):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    draws_t = tr.draw(R, NC, NI, 1., dev)
    ms = {}
    for tag, exact, split in (("exact", True, False), ("fused_split", False, True), ("fused", False, False)):
        tr.exact, tr.fused_split = exact, split
        ms[tag] = timed(step)
        ms[tag + "_forward"] = timed(lambda: tr.forward(o, d, hist[:1], NC, NI, NEAR, FAR, draws_t[0], None, 0., draws_t[2]))
    mac_fwd = R * (NC * (MAC_COARSE + 128 * 128 + 64 * (128 + 27) + 64 * 3) + (NC + NI) * MAC_FINE)
    # what the fused step moves through HBM: every layer input X_l and every pre-activation gradient G_l, written once by the chains
    # and read once by the weight-gradient stream — the fine network's as ONE f16 plane (2 bytes per element), the coarse network's
    # as hi | lo planes (4 bytes): csrc/nerfh_fused_train.h planes_of()
    wt_f, wt_c = -(-R * (NC + NI) // 256) * 8, -(-R * NC // 256) * 8
    stored = wt_f * (96 + 98) * 1024 + wt_c * (80 + 80) * 2048
    tf = 6.0 * mac_fwd / (ms["fused"] * 1e-3) / 1e12
    # matrix instructions the fused step ISSUES (the model profiles/rNN_roofline.md's N1 section is checked against,
    # tests/test_host_logic.py): forward and data-gradient chains three f16 MFMAs per product in both networks; the weight-gradient stream
    # three for the coarse network (hi | lo planes) but ONE for the fine network (one stored f16 plane per operand)
    mac_c, mac_f = mac_fwd - R * (NC + NI) * MAC_FINE, R * (NC + NI) * MAC_FINE
    issued_tf = 2.0 * (9.0 * mac_c + 7.0 * mac_f) / (ms["fused"] * 1e-3) / 1e12
    return {"workload": "one NeRF-H optimisation step (run_nerf.py:32-80): 1536 random rays, 64+128 samples, netwidth 128, perturb 1: "
                        "training-mode render, NerfWLoss, gradients of all 64 parameter tensors, Adam",
            "step_ms": ms["fused"], "forward_ms": ms["fused_forward"], "rays_per_s": R / ms["fused"] * 1e3,
            "arithmetic": "split-f16 MFMA (three v_mfma_f32_32x32x16_f16 per product, fp32 accumulation: fp32-grade), register-resident forward "
                          "and data-gradient chains, weight gradients streamed over the stored operands (csrc/nerfh_fused_*.hip)",
            "algorithmic_TFLOPs": tf, "f16_mfma_frac_of_nominal": 3.0 * tf / PEAK_TFLOPS["f16"], "frac_of_sustained_mfma": of_sustained(3.0 * tf),
            "f16_mfma_issued_frac_of_nominal": issued_tf / PEAK_TFLOPS["f16"],
            "issued_note": "f16_mfma_frac_of_nominal prices every product at three MFMAs (the split-f16 convention); the one-plane fine "
                           "weight-gradient stream issues one, so the instructions actually issued amount to f16_mfma_issued_frac_of_nominal "
                           "(agrees with SQ_INSTS_MFMA per step in profiles/*_train_step_pmc.json)",
            "flops_note": "forward 2 x MAC, data gradients 2 x MAC, weight gradients 2 x MAC; x 3 f16 MFMAs per product against the 2.5 PFLOP/s peak",
            "stored_operand_bytes_per_step": stored,
            "fused_split_step": {"step_ms": ms["fused_split"], "stored_operand_bytes_per_step": (wt_f * (96 + 98) + wt_c * (80 + 80)) * 2048,
                                 "what": "DFN_TRAIN_FUSED_SPLIT: the fine network's stored operands as hi | lo planes too (the round-4 layout)"},
            "hbm_GBps_floor_over_the_step": 2.0 * stored / (ms["fused"] * 1e-3) / 1e9,
            "hbm_note": "X_l and G_l written once (chains) and read once (weight-gradient stream): 2 x stored bytes over the WHOLE step time — "
                        "a floor; per kernel: profiles/r06_train_step_kernel_stats.csv.  Fine network: one f16 plane per stored operand "
                        "(round 4: hi | lo, 4.67 GB per step), coarse network: hi | lo",
            "range_check": "NerfHTrainer.range_check = 'skip': the range flag is read without draining the stream; a flagged step leaves zero gradients",
            "range_flags_after_the_steps": range_flags,
            "exact_fp32_step": {"step_ms": ms["exact"], "forward_ms": ms["exact_forward"], "arithmetic": "exact fp32 MFMA (v_mfma_f32_32x32x2_f32), layer by layer, "
                                "activations in HBM (DFN_TRAIN_EXACT; any netwidth)",
                                "fp32_mfma_frac": 6.0 * mac_fwd / (ms["exact"] * 1e-3) / 1e12 / PEAK_TFLOPS["f32"]},
            "parity_256_rays_vs_oracle_autograd": {"fused": par["fused"], "fused_split": par["fused_split"], "exact": par["exact"], "raw_noise_std": 1.0, "cpu_oracle_step_s": cpu_s}}


def secondary_trained_weights(dev):
    """Parity on TRAINED-LIKE weights (random-init weights are contractive, trained checkpoints amplify error): NeRF-H
    trained natively on a synthetic scene with real occupancy (tests/golden/trained_nerfh_weights.npz, tools/gpu_train_scene.py); one
    held-out 60 x 80 frame at 64 + 128 in the three arithmetic modes against the CPU oracle (pinned to the reference on these very weights
    by G15) and against the analytic ground truth of the scene."""
    from dfnet_amd import engine as eng, synthetic as syn
    from oracle import nerfh_oracle as orc
    T = torch.from_numpy
    cw, fw, ea, et = syn.trained_nerfh_weights()
    E = eng.NerfHEngine().load_numpy(cw, fw, ea, et)
    Hh, Ww, focal = 60, 80, 585.0 / 8
    c2w = syn.orbit_pose(7, 16)
    gt = T(syn.analytic_scene_image(c2w[:3, :4], Hh, Ww, focal))
    t0 = time.perf_counter()
    with torch.no_grad():
        ref = orc.render(Hh, Ww, focal, 32768, {k: T(v) for k, v in cw.items()}, {k: T(v) for k, v in fw.items()}, T(ea), T(et), NC, NI,
                         NEAR, FAR, syn.HIST_IDX, c2w=T(c2w))
    cpu_s = time.perf_counter() - t0
    psnr = lambda a, b: float(-10. * torch.log10(((a - b) ** 2).mean().clamp_min(1e-30)))
    out = {"workload": "NeRF-H trained for 20 000 fused HIP steps on three shaded spheres before a checkered wall; held-out 60x80 frame, 64+128 "
                       "samples, test-time render_image", "oracle_psnr_vs_ground_truth_db": psnr(ref[0], gt), "cpu_oracle_s": cpu_s,
           "largest_weight": float(max(np.abs(v).max() for v in fw.values())), "modes": {}}
    E.range_flags()
    for prec in ("f16x3", "f32", "f16"):
        rgb, disp, acc = E.render_image(T(c2w).to(dev), Hh, Ww, focal, T(syn.HIST_IDX).to(dev), NC, NI, NEAR, FAR, precision=prec)
        torch.cuda.synchronize()
        rgb, disp = rgb.cpu(), disp.cpu()
        err = (rgb - ref[0]).abs()
        out["modes"][prec] = {"psnr_vs_oracle_db": psnr(rgb, ref[0]), "psnr_vs_ground_truth_db": psnr(rgb, gt),
                              "rgb_max_rel_vs_oracle": float(err.max() / ref[0].abs().max()), "rgb_median_abs_vs_oracle": float(err.median()),
                              "disp_max_rel_vs_oracle": float((disp - ref[1]).abs().max() / ref[1].abs().max()), "range_flags": E.range_flags()}
    out["note"] = ("worst-pixel differences on trained weights are conditioning, not arithmetic: the reference's own fp32 sits 1e-2 from a float64 "
                   "evaluation at surface-grazing pixels (tests/test_gpu_nerfh.py::test_trained_weights_render_vs_reference measures all modes "
                   "against that float64 yardstick; ..._stages_on_the_references_own_samples holds network + compositor to 2e-5 on the "
                   "reference's own z_vals)")
    # One 640 x 480 frame of the trained scene (BASELINE configs[1]'s size), every arithmetic mode rendered in full on the device; the CPU
    # oracle — fp32 (= the reference, G15) and float64 (the yardstick) — on the 4 x 4 pixel lattice of it (19 200 rays: rays are
    # independent, so the lattice pixels of the full frame ARE the render of those rays).  Per mode: the fraction of lattice pixels further
    # than north_star's 1e-3 (of the map's range) from the fp32 oracle and from float64, beside the same fraction for the fp32 oracle itself.
    c2w_f = syn.orbit_pose(5, 16)
    ro, rd = orc.get_rays(H, W, FOCAL, T(c2w_f)[:3, :4])
    ro, rd = ro[::4, ::4].reshape(-1, 3), rd[::4, ::4].reshape(-1, 3)
    rows = orc.pack_ray_rows(ro, rd, NEAR, FAR, syn.HIST_IDX)
    tt = lambda d, dt: {k: T(v).to(dt) for k, v in d.items()}
    t0 = time.perf_counter()
    with torch.no_grad():
        r32 = orc.render_rays(rows, tt(cw, torch.float32), tt(fw, torch.float32), T(ea), T(et), NC, NI)
        prev = torch.get_default_dtype()
        torch.set_default_dtype(torch.float64)
        try:
            r64 = orc.render_rays(rows.double(), tt(cw, torch.float64), tt(fw, torch.float64), T(ea).double(), T(et).double(), NC, NI)
        finally:
            torch.set_default_dtype(prev)
    lattice_s = time.perf_counter() - t0
    scale = {k: float(r64[k].abs().max()) for k in ("rgb_map", "disp_map")}

    def beyond(a, b, k):    # fraction of pixels with any channel further than 1e-3 of the map's range
        e = (a.double() - b.double()).abs() / scale[k]
        e = e.amax(-1) if e.dim() == 2 else e
        return float((e > 1e-3).double().mean()), float(e.max())
    frame = {"size": "640x480, 64+128, orbit pose 5/16 of the trained scene", "oracle_pixels": int(rows.shape[0]), "cpu_oracle_fp32_and_fp64_s": lattice_s,
             "reference_fp32_vs_float64": {k: dict(zip(("frac_beyond_1e-3", "max_rel"), beyond(r32[k], r64[k], k))) for k in scale}, "modes": {}}
    pose_d, hist_d = T(c2w_f).to(dev), T(syn.HIST_IDX).to(dev)
    for tag, prec, c16 in (("f16x3", "f16x3", False), ("f32", "f32", False), ("f16x3_fine_f16_coarse", "f16x3", True), ("f16", "f16", False)):
        E.set_render_options(coarse_f16=c16)
        try:
            rgb, disp, _ = E.render_image(pose_d, H, W, FOCAL, hist_d, NC, NI, NEAR, FAR, precision=prec)
            torch.cuda.synchronize()
        finally:
            E.set_render_options(coarse_f16=False)
        got = {"rgb_map": rgb[::4, ::4].reshape(-1, 3).cpu(), "disp_map": disp[::4, ::4].reshape(-1).cpu()}
        frame["modes"][tag] = {"range_flags": E.range_flags()}
        for k in scale:
            fr, mx = beyond(got[k], r32[k], k)
            fr64, mx64 = beyond(got[k], r64[k], k)
            frame["modes"][tag][k] = {"frac_beyond_1e-3_vs_reference_fp32": fr, "max_rel_vs_reference_fp32": mx,
                                      "frac_beyond_1e-3_vs_float64": fr64, "max_rel_vs_float64": mx64}
    out["frame_640x480"] = frame
    return out


def secondary_w256(dev):
    """SURVEY §8(d) 'also report' netwidth 256 (325.9 MFLOP per ray, 100.1 TFLOP per 640x480 frame) on the register-resident
    netwidth-256 kernels (f16 / split-f16 / exact fp32) and on the generic-width path, each with its parity against the oracle."""
    from dfnet_amd import engine as eng, synthetic as syn
    from oracle import nerfh_oracle as orc
    T = torch.from_numpy
    cw, fw, ea, et = syn.nerfh_weights(0, W=256)
    E = eng.NerfHEngine(width=256).load_numpy(cw, fw, ea, et)
    pose, hist = T(syn.orbit_pose(0, 8)).to(dev), T(syn.HIST_IDX).to(dev)
    ro, rd = orc.get_rays(H, W, FOCAL, T(syn.orbit_pose(0, 8))[:3, :4])
    sel = torch.randperm(H * W, generator=torch.Generator().manual_seed(0))[:512]
    rows = orc.pack_ray_rows(ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel], NEAR, FAR, syn.HIST_IDX)
    with torch.no_grad():
        ref = orc.render_rays(rows, {k: T(v) for k, v in cw.items()}, {k: T(v) for k, v in fw.items()}, T(ea), T(et), NC, NI)
    flops = 325.9e6 * H * W
    out = {"workload": "netwidth 256 NeRF-H, 640x480 frames at 64+128 (100.1 TFLOP algorithmic per frame)", "precisions": {}}
    for prec, reps in (("f16", 3), ("f16x3", 2), ("f32", 1), ("generic", 1)):
        E.render_image(pose, 120, 160, FOCAL / 4, hist, NC, NI, NEAR, FAR, precision=prec)   # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            E.render_image(pose, H, W, FOCAL, hist, NC, NI, NEAR, FAR, precision=prec)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        rgb, disp, _, _ = E.render_rays(rows[:, 0:3].to(dev), rows[:, 3:6].to(dev), hist, NC, NI, NEAR, FAR, precision=prec)
        peak = {"f16": PEAK_TFLOPS["f16"], "f16x3": PEAK_TFLOPS["f16x3"], "f32": PEAK_TFLOPS["f32"], "generic": PEAK_TFLOPS["f32"]}[prec]
        out["precisions"][prec] = {
            "value": H * W / dt, "unit": "rays/s", "ms_per_frame": dt * 1e3, "algorithmic_TFLOPs": flops / dt / 1e12,
            "mfma_frac": flops / dt / 1e12 / peak,
            "arithmetic": PREC_TEXT.get(prec, "generic-width path: layer by layer, exact fp32 MFMA, activations in HBM"),
            "parity_vs_oracle": {"rgb_max_rel": float((rgb.cpu() - ref["rgb_map"]).abs().max() / ref["rgb_map"].abs().max()),
                                 "disp_max_rel": float((disp.cpu() - ref["disp_map"]).abs().max() / ref["disp_map"].abs().max()), "rays": 512}}
    out["value"], out["unit"] = out["precisions"]["f16"]["value"], "rays/s"
    return out


# ---------------------------------------------------------------------------------------------- launch
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N ranks of this same file (one per GPU) on 127.0.0.1."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


def dry_main(args, rank, world):
    """--cpu-dry: the launch / sharding / gather / timing / JSON skeleton of the N-rank run with the HIP render replaced by a
    deterministic frame fill — what tests/test_host_logic.py runs at world size 2 over gloo (no GPU here)."""
    from dfnet_amd import dist as ddist
    dev = torch.device("cpu")
    K = args.steps
    n_frames = K * world
    lo, hi = ddist.frame_block(n_frames, rank, world)
    h, w = 6, 8
    outs, (rgbs, disps) = ddist.root_buffers([(h, w, 3), (h, w)], n_frames, dev)
    ddist.barrier()
    t0 = time.perf_counter()
    for k in range(K):   # the "render": frame lo + k filled with its global index
        rgbs[k].fill_(float(lo + k))
        disps[k].fill_(float(lo + k) + .5)
    t_render = time.perf_counter() - t0
    (all_rgb, all_disp), _ = ddist.gather_frames_direct([rgbs, disps], n_frames, outs=outs)
    t_gather = time.perf_counter() - t0 - t_render
    ddist.barrier()
    dt = ddist.max_over_ranks(time.perf_counter() - t0, dev)
    stats = ddist.all_gather_floats([t_render, t_gather, float("nan"), float("nan"), 0.], dev)
    if rank == 0:
        ok = bool((all_rgb[:, 0, 0, 0] == torch.arange(n_frames, dtype=torch.float32)).all()) and \
            bool((all_disp[:, 0, 0] == torch.arange(n_frames, dtype=torch.float32) + .5).all())
        in_place = world == 1 or all_rgb.data_ptr() == outs[0].data_ptr()
        print(json.dumps({"metric": "dry run (no GPU work)", "value": world * K * h * w / dt, "unit": "rays/s", "n_gpus": world,
                          "steps": K, "warmup": args.warmup, "scaling": "weak", "frames_gathered_in_order": ok,
                          "received_in_place": bool(in_place), "ranks": per_rank_record(stats, ddist.gathered_bytes([rgbs, disps], n_frames)),
                          "data": "synthetic", "dtype": "none"}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--precision", default="f16x3", choices=["f16", "f32", "f16x3"],
                    help="arithmetic of the headline value (default: split-f16, fp32-grade = the reference's precision)")
    ap.add_argument("--cpu-sample", type=int, default=16384, help="rays in the CPU baseline subset (0 = skip the oracle legs)")
    ap.add_argument("--no-extras", action="store_true", help="headline only (no precisions / hbm / secondary records)")
    ap.add_argument("--backend", default=None, choices=["nccl", "gloo"], help="torch.distributed backend (default nccl = RCCL)")
    ap.add_argument("--cpu-dry", action="store_true", help="no GPU work: exercise launch + sharding + gather only (CPU tests)")
    ap.add_argument("--cpu-worker", nargs=3, metavar=("MODE", "RAYS", "OUT"), help="internal: the pinned child process of the CPU baseline")
    args = ap.parse_args()
    if args.cpu_worker:
        cpu_baseline_worker(args.cpu_worker[0], int(args.cpu_worker[1]), args.cpu_worker[2])
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    from dfnet_amd import dist as ddist
    rank, world, local = ddist.init_from_env(backend=args.backend or ("gloo" if args.cpu_dry else "nccl"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.cpu_dry:
        dry_main(args, rank, world)
        if world > 1:
            torch.distributed.destroy_process_group()
        return

    from dfnet_amd import _lib, engine as eng, synthetic as syn
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cw, fw, ea, et = syn.nerfh_weights(0)
    E = eng.NerfHEngine(precision=args.precision).load_numpy(cw, fw, ea, et)
    lib = _lib.load()
    K, Wm = args.steps, args.warmup
    n_frames = K * world
    lo, _ = ddist.frame_block(n_frames, rank, world)
    poses = torch.stack([torch.from_numpy(syn.orbit_pose(lo + k, n_frames)) for k in range(K)]).to(dev)
    hist = torch.from_numpy(syn.HIST_IDX).to(dev)
    # rank 0 renders into its block of the final [n_frames, ...] tensors; the gather receives the peers' blocks in place
    outs, (rgbs, disps) = ddist.root_buffers([(H, W, 3), (H, W)], n_frames, dev)
    acc = torch.empty(H, W, device=dev)

    def gather():
        ddist.gather_frames_direct([rgbs, disps], n_frames, outs=outs)   # one grouped exchange: every peer sends rgb + disp to rank 0

    dt, prof, stats = timed_render(E, lib, args.precision, poses, hist, rgbs, disps, acc, K, Wm, world, gather)

    if rank == 0:
        rays = H * W
        value = world * K * rays / dt
        roof = mlp_roofline(prof, K, args.precision, value / world)
        roof["traffic"] = pmc_traffic("nerfh_fine_kernel", args.precision)
        roof["traffic_note"] = ("HBM bytes per launch of this precision's instantiation, rocprofv3 PMC (2*FETCH_SIZE + WRITE_SIZE) from "
                                "profiles/*_pmc_summary*.json; algorithmic bytes per launch = rays * (768 z + 24 o,d + 512 ray-bias + "
                                "segment composites: 144 at 64-sample segments (f16), 288 at 32-sample segments (f16x3 / f32)); raw "
                                "(6912 B/ray) stays in registers since compositing is fused")
        line = {
            "metric": "rendered rays/sec (64+128 samples, 640x480)",
            "value": value, "unit": "rays/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: synthetic random-weight NeRF-H (D=8, W=128), 640x480, "
                                   "64+128 samples, test-time render_image, 1 frame per step per GPU",
                       "rays_per_step_per_gpu": rays, "precision": PREC_TEXT[args.precision],
                       "precision_gate": PREC_GATE[args.precision],
                       "parallelism": f"frames sharded over {world} GPU(s), gather at end",
                       "collectives": ("one grouped RCCL send / receive batch of rgb + disp into rank 0's final tensors, inside the timed region" + (" (forced at world size 1: padded gather)" if world == 1 else ""))
                                      if ddist.active() else "none (one rank)"},
            "roofline": roof,
            "ranks": per_rank_record(stats, ddist.gathered_bytes([rgbs, disps], n_frames)),
        }
        if world == 1 and not args.no_extras:
            sus = sustained_mfma(lib)
            SUSTAINED.update(sus)
            roof["sustained_mfma"] = sus
            add_sustained(roof, sus, args.precision)
            line["hbm"] = hbm_records(prof, K, E, dev)
            ref_pack = None
            if args.cpu_sample > 0:
                rec, ref_pack = cpu_baseline(args.cpu_sample)
                rec["parity_vs_oracle"] = parity(E, ref_pack[0], ref_pack[1], args.precision, dev)
                line["cpu_baseline"] = rec
            precs = {}
            if args.precision == "f16x3":
                line["config"]["fp32_grade_check"] = fp32_grade_check(E, dev)
            for prec, k2, w2 in (("f16x3", max(2, min(K, 6)), 1), ("f32", 2, 1), ("f16", max(2, min(K, 10)), 2)):
                if prec == args.precision:
                    continue
                dt2, prof2, _ = timed_render(E, lib, prec, poses, hist, rgbs, disps, acc, k2, w2)
                v2 = k2 * rays / dt2
                precs[prec] = {"value": v2, "unit": "rays/s", "ms_per_step": dt2 / k2 * 1e3, "steps": k2, "warmup": w2,
                               "arithmetic": PREC_TEXT[prec], "roofline": mlp_roofline(prof2, k2, prec, v2)}
                precs[prec]["roofline"]["traffic"] = pmc_traffic("nerfh_fine_kernel", prec)
                add_sustained(precs[prec]["roofline"], sus, prec)
                if ref_pack is not None:
                    precs[prec]["parity_vs_oracle"] = parity(E, ref_pack[0], ref_pack[1], prec, dev)
            # split-f16 fine network with the coarse network in f16 (DFN_RENDER_COARSE_F16, `--coarse_precision f16`): an OPT-IN since
            # round 6 — on trained weights f16 densities move importance samples across surfaces (secondary.nerfh_trained_weights)
            if args.precision != "f16":
                E.set_render_options(coarse_f16=True)
                try:
                    k2 = max(2, min(K, 6))
                    dt2, prof2, _ = timed_render(E, lib, "f16x3", poses, hist, rgbs, disps, acc, k2, 1)
                    v2 = k2 * rays / dt2
                    precs["f16x3_fine_f16_coarse"] = {
                        "value": v2, "unit": "rays/s", "ms_per_step": dt2 / k2 * 1e3, "steps": k2, "warmup": 1,
                        "arithmetic": "fine network split-f16 (fp32-grade), coarse network (sample placement only) f16 MFMA inputs",
                        "roofline": mlp_roofline(prof2, k2, "f16x3", v2)}
                    precs["f16x3_fine_f16_coarse"]["roofline"].pop("whole_path_mfma_frac", None)   # two peaks on this path: not defined
                    add_sustained(precs["f16x3_fine_f16_coarse"]["roofline"], sus, "f16x3")
                    if ref_pack is not None:
                        precs["f16x3_fine_f16_coarse"]["parity_vs_oracle"] = parity(E, ref_pack[0], ref_pack[1], "f16x3", dev)
                finally:
                    E.set_render_options(coarse_f16=False)
            line["precisions"] = precs
            sec = {}
            if args.cpu_sample > 0:
                torch.set_num_threads(int(line["cpu_baseline"]["cores"]))   # the oracle legs below: the fastest thread count found
            for name, fn in (("dfnet_forward_c4", secondary_dfnet), ("dfnet_dm_step_c5", secondary_dm_step),
                             ("dfnet_train_step_n2", secondary_dfnet_train), ("nerfh_train_step_n1", secondary_nerfh_train), ("nerfh_netwidth_256", secondary_w256),
                             ("nerfh_trained_weights", secondary_trained_weights)):
                try:
                    sec[name] = fn(dev) if args.cpu_sample > 0 else {"skipped": "--cpu-sample 0"}
                except Exception as e:  # a failing secondary must not lose the headline line
                    sec[name] = {"error": f"{type(e).__name__}: {e}"}
            line["secondary"] = sec
        elif world == 1 and args.cpu_sample > 0:
            rec, ref_pack = cpu_baseline(args.cpu_sample)
            rec["parity_vs_oracle"] = parity(E, ref_pack[0], ref_pack[1], args.precision, dev)
            line["cpu_baseline"] = rec
        print(json.dumps(line), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
