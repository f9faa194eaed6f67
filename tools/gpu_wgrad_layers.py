#!/usr/bin/env python3
"""Per-layer time of the split-storage conv weight gradient (csrc/dfnet_wgrad_s.hip) at the DFNet_dm step's shapes (batch 4,
240x320 frames) — or `B H W` from the command line.
  run    : calls dfn_conv_wgrad REPS times per layer (under `rocprofv3 --kernel-trace --output-format csv -d DIR -o w`)
  report : DIR -> per layer: mean kernel time of the stream + finalize launches, algorithmic f16-MFMA FLOPs (x3 split), fraction
           of the 2.5 PFLOP/s nominal peak."""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REPS = 4


def layers(B, H, W):
    out = []
    h, w, cin = H, W, 64
    cfg = [(64, 1), (128, 2), (128, 2), (256, 3), (256, 3), (256, 3), (512, 4), (512, 4), (512, 4), (512, 5), (512, 5), (512, 5)]
    names = ["conv1_2", "conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv3_3", "conv4_1", "conv4_2", "conv4_3", "conv5_1", "conv5_2", "conv5_3"]
    stage = 1
    for (cout, st), nm in zip(cfg, names):
        if st != stage:
            h, w, stage = h // 2, w // 2, st
        out.append((nm, B, h, w, cout, cin, 3))
        cin = cout
    out.append(("adapt0_5x5", B, H, W, 128, 64, 5))
    out.append(("adapt0_1x1", B, H, W, 64, 64, 1))
    return out


def main():
    mode = sys.argv[1]
    B, H, W = (int(v) for v in sys.argv[3:6]) if len(sys.argv) > 5 else (4, 240, 320)
    L = layers(B, H, W)
    if mode == "run":
        import ctypes, torch
        from dfnet_amd import _lib
        from dfnet_amd._lib import check, current_stream, ptr
        lib = _lib.load()
        dev = torch.device("cuda:0")
        for nm, b, h, w, cout, cin, ks in L:
            g = torch.randn(b, cout, h, w, device=dev)
            x = torch.relu(torch.randn(b, cin, h, w, device=dev))
            nb = lib.dfn_conv_wgrad_scratch_bytes(b, h, w, cout, cin, ks)
            scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
            dW, db = torch.empty(cout, cin, ks, ks, device=dev), torch.empty(cout, device=dev)
            for _ in range(REPS):
                check(lib.dfn_conv_wgrad(ptr(g), ptr(x), b, h, w, cout, cin, ks, ptr(dW), ptr(db), ctypes.c_void_p(scratch.data_ptr()),
                                         scratch.numel(), current_stream()), nm)
            torch.cuda.synchronize()
        return
    rows = []
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    stream = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "conv_wgrad_s_kernel" in r["Kernel_Name"]]
    fin = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "wgrad_s_finalize_kernel" in r["Kernel_Name"]]
    i = j = 0
    tot = 0.0
    print(f"{'layer':12s} {'shape':>22s} {'stream us':>10s} {'finalize us':>11s} {'TFLOP/s x3':>10s} {'frac':>6s}")
    for nm, b, h, w, cout, cin, ks in L:
        per = 3 if ks == 5 else 1   # 5x5: kernel rows (0, 1), (2, 3), (4)
        s = stream[i:i + REPS * per]; i += REPS * per
        f = fin[j:j + REPS]; j += REPS
        if len(s) < REPS * per or len(f) < REPS:
            break
        ts = sum(s[per:]) / (REPS - 1)       # first repetition = warm-up
        tf = sum(f[1:]) / (REPS - 1)
        flop = 2.0 * b * h * w * cout * cin * ks * ks * 3
        tot += ts + tf
        print(f"{nm:12s} {f'{b}x{h}x{w} {cin}->{cout}':>22s} {ts:10.1f} {tf:11.1f} {flop / ts / 1e6:10.0f} {flop / ts / 1e6 / 2500:6.3f}")
    print(f"total {tot:.1f} us per step (3x3 encoder layers + level-0 adaptation layers)")


if __name__ == "__main__":
    main()
