"""Development: the backward blend's batches by entry count (a library built with -DSOAR_BWD_HIST):
python scripts/bwd_hist.py soar_amd/_lib/variants/bwd_hist.so [bench.py arguments]"""
import ctypes, os, runpy, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import soar_amd.hip_lib as h
h.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
try:
    runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bench.py"), run_name="__main__")
except SystemExit:
    pass
out = (ctypes.c_ulonglong * 16)()
assert h.lib().soar_debug_bwd_hist(out) == 0
names = ["1-8", "9-16", "17-32", "33-63", "64"]
tot_it = sum(out[:5]); tot_b = sum(out[8:13])
for i, n in enumerate(names):
    print("entries %-6s  batches %10d (%5.1f %%)  pixel iterations %11d (%5.1f %%)  mean pixels per batch %.1f" %
          (n, out[8 + i], 100.0 * out[8 + i] / max(tot_b, 1), out[i], 100.0 * out[i] / max(tot_it, 1), out[i] / max(out[8 + i], 1)), file=sys.stderr)
