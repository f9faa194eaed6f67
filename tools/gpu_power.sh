#!/bin/bash
# Socket power and shader clock (rocm-smi) while the render loop / the bare MFMA probe run: shows which workloads sit at the power cap.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
sample() {  # $1 = label, background pid in $2
  sleep 2.5
  for i in 1 2 3 4; do
    p=$(rocm-smi --showpower 2>/dev/null | grep -i "power" | grep -o "[0-9.]* *$" | head -1)
    c=$(rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | grep -o "([0-9]*Mhz)" | head -1)
    echo "$1: power $p W  sclk $c"
    sleep 0.7
  done
  wait $2
}
rocm-smi --showmaxpower 2>/dev/null | grep -i "max" | head -2
for prec in f16x3 f32 f16; do
  python - <<PY &
import sys, time, torch
sys.path.insert(0, "$R")
from dfnet_amd import engine as eng, synthetic as syn
cw, fw, ea, et = syn.nerfh_weights(0)
E = eng.NerfHEngine(precision="$prec").load_numpy(cw, fw, ea, et)
dev = "cuda:0"; hist = torch.from_numpy(syn.HIST_IDX).to(dev); pose = torch.from_numpy(syn.orbit_pose(0, 8)).to(dev)
t0 = time.time(); n = 0
while time.time() - t0 < 7.0:
    E.render_image(pose, 480, 640, 585.0, hist, 64, 128, 0.0, 2.5, precision="$prec"); n += 1
    if n % 4 == 0: torch.cuda.synchronize()
torch.cuda.synchronize()
print("$prec frames", n, "ms/frame", (time.time() - t0) / n * 1e3)
PY
  sample "render $prec" $!
done
for rnd in 1 0; do
  python - <<PY &
import sys, ctypes
sys.path.insert(0, "$R")
import bench
lib = bench.load_probe(); tf = ctypes.c_double()
lib.dfn_probe_mfma_rate($rnd, 4.5, ctypes.byref(tf), None); lib.dfn_probe_mfma_rate($rnd, 4.5, ctypes.byref(tf), None)
print("probe random=$rnd TFLOP/s", tf.value)
PY
  sample "mfma probe random=$rnd" $!
done
