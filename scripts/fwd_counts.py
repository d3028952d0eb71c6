"""Development: what the forward blend's two culling levels keep of a C3 frame's tile lists (variant built with -DSOAR_FWD_COUNT).
    python scripts/variant.py fwd_count rast_render_fwd.hip -DSOAR_FWD_COUNT && python scripts/fwd_counts.py soar_amd/_lib/variants/fwd_count.so"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import soar_amd.hip_lib as h
h.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
import bench
dev = torch.device("cuda:0")
seq, pool, _ = bench.build_sequence(sys.argv[2] if len(sys.argv) > 2 else "C3", dev)
bg = torch.tensor([0.2, 0.5, 0.7], device=dev)
L = h.lib()
out = (C.c_ulonglong * 4)()
with torch.no_grad():
    seq.render_frame(40, bg, with_occ=True)
    torch.cuda.synchronize()
    L.soar_debug_fwd_counts(out, 1)
    seq.render_frame(41, bg, with_occ=True)
    torch.cuda.synchronize()
    L.soar_debug_fwd_counts(out, 1)
chunks, entries, nq, ntodo = [int(v) for v in out]
print(f"(live wavefront, chunk) pairs {chunks}; entries per chunk {entries / chunks:.1f}; the quad keeps {nq / chunks:.1f} ({100 * nq / entries:.1f} %), "
      f"the block {ntodo / chunks:.2f} ({100 * ntodo / entries:.2f} % of the chunk, {100 * ntodo / max(nq, 1):.1f} % of the quad's)")
