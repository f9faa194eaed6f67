"""The multi-GPU path on the one GPU a test box has: RCCL is brought up through torch.distributed's "nccl" backend
under the same launcher the driver uses, and DFN_FORCE_COLLECTIVES=1 makes dfnet_amd.dist run its gather / all-reduce / barrier in
the group of one rank — communicator initialisation, device buffers and frame order are exercised for real; only the number of
peers is not.  (1 -> 8 scaling itself is the driver's to measure.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import sys, torch
sys.path.insert(0, sys.argv[1])
from dfnet_amd import dist as ddist
rank, world, local = ddist.init_from_env(backend="nccl")
assert torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl" and world == 1 and ddist.active()
dev = torch.device("cuda", local)
frames = torch.arange(5, dtype=torch.float32, device=dev)[:, None, None, None].expand(5, 6, 8, 3).contiguous() + 0.25
out = ddist.gather_frames(frames, 5)                      # a real RCCL gather (not the world-1 early-out)
assert out is not frames and out.shape == frames.shape and torch.equal(out, frames)
# the direct form of render_path / bench.py in the one-rank group (falls back to the padded gather: no peer to send to) + its record
outs, (v_rgb, v_err) = ddist.root_buffers([(6, 8, 3), ()], 5, dev)
v_rgb.copy_(frames); v_err.copy_(torch.arange(5, device=dev, dtype=torch.float32))
(d_rgb, d_err), extra = ddist.gather_frames_direct([v_rgb, v_err], 5, outs=outs, extra=torch.tensor([2.0], device=dev))
assert torch.equal(d_rgb, frames) and torch.equal(d_err, v_err) and extra.shape == (1, 1) and float(extra[0, 0]) == 2.0
assert ddist.all_gather_floats([1.0, 2.0], dev).tolist() == [[1.0, 2.0]]
p = torch.nn.Parameter(torch.ones(1 << 20, device=dev)); p.grad = torch.full_like(p, 3.0)
ddist.allreduce_gradients([p]); assert float(p.grad.mean()) == 3.0
ddist.allreduce_gradients([p], contributors=1); assert float(p.grad.mean()) == 3.0
assert ddist.max_over_ranks(1.5, dev) == 1.5
bn = torch.nn.BatchNorm2d(4).to(dev); bn.running_var.fill_(2.0); ddist.sync_buffers(bn); assert float(bn.running_var[0]) == 2.0
ddist.barrier(); torch.cuda.synchronize()
print("RCCL_WORLD1_OK", torch.cuda.nccl.version())
torch.distributed.destroy_process_group()
'''


def _launch(args, timeout=600):
    env = dict(os.environ, DFN_FORCE_COLLECTIVES="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                           "--master-port", "29655"] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def test_rccl_collectives_forced_at_world1(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    r = _launch([str(script), ROOT])
    assert r.returncode == 0, r.stderr[-3000:]
    assert "RCCL_WORLD1_OK" in r.stdout


def test_bench_under_torchrun_with_rccl_gather(tmp_path):
    """bench.py exactly as the driver launches it for N > 1 (torch.distributed.run, nccl), here with one rank: the frames go
    through the RCCL gather inside the timed region."""
    r = _launch([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--cpu-sample", "0", "--no-extras"])
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["value"] > 1e5
    assert "forced" in line["config"]["collectives"] and line["dtype"] == "f16x3"
    # the per-rank record that will explain the N-GPU runs: render / gather seconds, package power and shader clock over the timed region
    ranks = line["ranks"]
    assert len(ranks["render_s"]["per_rank"]) == 1 and ranks["render_s"]["max"] > 0 and ranks["gather_s"]["max"] >= 0
    assert ranks["gathered_bytes"] == 0   # one rank: every frame is the root's own
    assert ranks["samples_per_rank"][0] > 0 and 100 < ranks["power_w"]["mean"] < 1600 and 300 < ranks["sclk_mhz"]["mean"] < 2600, ranks
