"""How long after the last of four side-stream graphs has finished does the caller's stream start its next kernel?
(A) torch events (hipEventRecord / hipStreamWaitEvent), (B) stream memory operations: hipStreamWriteValue32 behind every graph,
hipStreamWaitValue32 on the caller's stream.  Device wall-clock stamps (soar_prof_timestamp) inside the graphs and behind the join."""
import ctypes as C, os, sys
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from soar_amd import hip_lib

L = hip_lib.lib()
hip = C.CDLL("libamdhip64.so")
dev = torch.device("cuda:0")
CAP = 8192
ring = torch.zeros((1 + 2 * CAP,), dtype=torch.int64, device=dev)
def stamp(tag, stream):
    assert L.soar_prof_timestamp(C.c_void_p(ring.data_ptr()), CAP, tag, C.c_void_p(stream)) == 0
n = 4
streams = [torch.cuda.Stream(device=dev) for _ in range(n)]
x = [torch.randn(2048, 2048, device=dev) for _ in range(n)]
for i in range(n):
    (x[i] @ x[i]).sum().item()
graphs = []
cap = torch.cuda.Stream(device=dev)
for i in range(n):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=cap):
        y = x[i] @ x[i]
        y = y @ x[i]
        stamp(10 + i, cap.cuda_stream)
    graphs.append(g)
torch.cuda.synchronize()
sig = []
rc = 0
for i in range(n):
    f = C.c_void_p()
    rc |= hip.hipExtMallocWithFlags(C.byref(f), C.c_size_t(8), C.c_uint(0x2))      # hipMallocSignalMemory: 8 bytes each
    sig.append(f.value)
print("hipExtMallocWithFlags(signal memory) rc", rc, sig)
if rc:
    hip.hipGetLastError()
plain = torch.zeros(n, dtype=torch.int32, device=dev)
hip.hipStreamWriteValue32.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint]
hip.hipStreamWaitValue32.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint, C.c_uint32]

def run(method, step, ptrs):
    main = torch.cuda.current_stream(dev)
    ev = torch.cuda.Event(); ev.record(main)
    done = []
    for i in range(n):
        s = streams[i]
        s.wait_event(ev)
        with torch.cuda.stream(s):
            graphs[i].replay()
        if method == "events":
            e = torch.cuda.Event(); e.record(s); done.append(e)
        else:
            rc = hip.hipStreamWriteValue32(C.c_void_p(s.cuda_stream), C.c_void_p(ptrs[i]), C.c_uint32(step), 0)
            assert rc == 0, rc
    if method == "events":
        for e in done:
            main.wait_event(e)
    else:
        for i in range(n):
            rc = hip.hipStreamWaitValue32(C.c_void_p(main.cuda_stream), C.c_void_p(ptrs[i]), C.c_uint32(step), 0, 0xFFFFFFFF)   # 0: >=
            assert rc == 0, rc
    stamp(99, main.cuda_stream)

for method, ptr in (("events", [0]), ("values(signal memory)", sig if rc == 0 else None), ("values(plain memory)", [plain.data_ptr() + 4 * i for i in range(n)])):
    if ptr is None:
        continue
    try:
        for step in range(1, 6):
            run(method.split("(")[0], step, ptr)
        torch.cuda.synchronize()
        ring.zero_()
        for step in range(6, 46):
            run(method.split("(")[0], step, ptr)
        torch.cuda.synchronize()
    except AssertionError as e:
        print(method, "failed: hip error", e)
        continue
    r = ring.cpu().numpy()
    cnt = int(r[0]); ev = r[1:1 + 2 * cnt].reshape(cnt, 2)
    tags, clk = ev[:, 0], ev[:, 1] / 100.0
    joins = clk[tags == 99]
    lat = []
    for j in joins:
        ends = [clk[(tags == 10 + i) & (clk <= j)].max() for i in range(n)]
        lat.append(j - max(ends))
    print(f"{method}: join latency after the last graph's last kernel: median {np.median(lat):.1f} us, min {np.min(lat):.1f}, max {np.max(lat):.1f}; step period {np.median(np.diff(joins)):.0f} us")
