"""Per-wave timeline of ONE forward blend launch at C3 (SOAR_WAVE_LOG diagnostic), fused occlusion pass on."""
import os, sys, subprocess
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["SOAR_WAVE_LOG"] = "/tmp/fwd_log.bin"
import torch
import bench
seq, targets, parts = bench.build_sequence("C3", torch.device("cuda:0"))  # targets: resident pool [sets,7,H,W]
bg = torch.tensor([0.2, 0.5, 0.7], device="cuda:0")
with torch.no_grad():
    seq.render_frames([0], bg, with_occ=(len(sys.argv) < 2 or sys.argv[1] != "noocc"))
torch.cuda.synchronize()
subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "wave_log.py"), "/tmp/fwd_log.bin"])
