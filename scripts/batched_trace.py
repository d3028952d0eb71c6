"""Mean duration of the batched launches (gridDim.y = frames per step) of one bench run, from a rocprofv3 kernel trace:
   rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python3 bench.py ... ; python scripts/batched_trace.py DIR [frames]"""
import collections, csv, glob, re, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
d = collections.defaultdict(list)
for r in rows:
    gy = int(r.get("Grid_Size_Y", "1") or 1) // max(int(r.get("Workgroup_Size_Y", "1") or 1), 1)
    if gy == n:
        d[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000)
tot = 0.0
steps = min(len(v) for v in d.values()) if d else 1            # (a kernel launched once per step)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]) / len(kv[1])):
    m = sum(v) / len(v)
    tot += m * len(v) / steps
    m_ = re.search(r"(\w+_kernel)", k)
    name = m_.group(1) if m_ else k[:60]
    print(f"{m:8.1f} us x{len(v):3d}  {name}")
print(f"sum over one step: {tot:.0f} us")
