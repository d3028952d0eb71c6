"""Host-side profile (cProfile) of the plugin training step of scripts/plugin_time.py: where the Python time of a frame goes."""
import cProfile, os, pstats, sys
os.environ["SOAR_PLUGIN_TIME_IMPORT_ONLY"] = "1"
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import plugin_time as pt
for f in range(pt.F + 4):
    pt.step(f, "fused")
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for f in range(60):
    pt.step(f, "fused")
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
