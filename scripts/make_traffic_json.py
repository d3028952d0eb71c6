"""Summarise the FETCH_SIZE / WRITE_SIZE passes of scripts/profile_round.sh into profiles/<tag>_hbm_traffic.json.
usage: python scripts/make_traffic_json.py gpurun_out/<tag> profiles/<tag>_hbm_traffic.json"""
import collections, csv, glob, json, os, re, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from soar_amd import build
src, dst = sys.argv[1], sys.argv[2]


def agg(pattern, name):
    d = collections.defaultdict(list)
    for path in glob.glob(pattern):
        for r in csv.DictReader(open(path)):
            m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
            if r["Counter_Name"] == name and m and "soar" in r["Kernel_Name"]:
                d[m.group(1)].append(float(r["Counter_Value"]))
    return d


f = agg(src + "/pmc_fetch/*/*counter_collection.csv", "FETCH_SIZE")
w = agg(src + "/pmc_write/*/*counter_collection.csv", "WRITE_SIZE")
# kernels whose reads are dominated by 16-byte-per-lane loads (64-byte records / float4 rows): FETCH_SIZE x2 on gfx950
wide = {"render_forward_kernel", "render_backward_slots_kernel", "render_backward_blocks_kernel", "render_backward_entries_kernel",
        "geometry_backward_kernel", "frame_loss_kernel", "block_mask_kernel"}
out = {"_about": "per-launch HBM-side traffic of the soar kernels at C3 (bench.py --steps 2 --warmup 1), rocprofv3 --pmc FETCH_SIZE "
                 "and --pmc WRITE_SIZE in separate passes (scripts/profile_round.sh). Counters are in KB. fetch_bytes applies the "
                 "gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE reports half of the bytes of 16-byte-per-lane reads) to the "
                 "kernels whose reads are dominated by 16-byte loads; for the others the raw value is kept (uncalibrated width).",
       "build_digest": build.source_digest(),      # the sources the profiled library was built from (bench.py checks it)
       "kernels": {}}
for k in sorted(f):
    fv = sum(f[k]) / len(f[k]) * 1024
    wv = sum(w.get(k, [0])) / max(len(w.get(k, [0])), 1) * 1024
    fb = fv * 2 if k in wide else fv
    out["kernels"][k] = {"launches_sampled": len(f[k]), "FETCH_SIZE_raw_bytes": round(fv), "fetch_bytes": round(fb),
                         "fetch_corrected_x2": k in wide, "WRITE_SIZE_bytes": round(wv), "traffic_bytes": round(fb + wv)}
json.dump(out, open(dst, "w"), indent=1)
for k, v in out["kernels"].items():
    print("%-32s %8.1f MB" % (k, v["traffic_bytes"] / 1e6))
