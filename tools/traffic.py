"""HBM traffic of the pass's kernels by PMC, per launch, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in
SEPARATE rocprofv3 --pmc passes (never with other trace domains), FETCH_SIZE doubled (gfx950 tallies 128-B read requests at
64 B), both in KiB; a third pass gives the L2 hit rate.  Writes one entry per workload into a JSON file that bench.py reads
(roofline.traffic is reported only while the kernel sources are the ones measured).

  python3 tools/traffic.py <out.json> <scene> <spp> <calls> <accel> [option=value ...]
"""
import csv, glob, hashlib, json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def sources_sha():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "computeraytracer_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")) or name in ("crt_api.cpp", "crt_bvh.cpp"):     # (as bench.py's kernel_sources_sha256)
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()


def main():
    out, scene, spp, calls, accel = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    opts = sys.argv[6:]
    env = dict(os.environ, TMPDIR="/tmp")
    agg, launches = {}, {}
    for grp in (["FETCH_SIZE"], ["WRITE_SIZE"], ["TCC_HIT_sum", "TCC_MISS_sum"]):
        d = tempfile.mkdtemp(prefix="pmc_", dir="/tmp")
        cmd = ["rocprofv3", "--pmc"] + grp + ["--kernel-trace", "--output-format", "csv", "-d", d, "--",
               "python3", os.path.join(ROOT, "tools", "prof_workload.py"), scene, str(spp), str(calls), accel] + opts
        p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
        print("pass", grp, "rc", p.returncode, flush=True)
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"].split("(")[0].replace("void crt::", "").replace("crt::", "")
                a = agg.setdefault(k, {})
                a[row["Counter_Name"]] = a.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                if row["Counter_Name"] == grp[0]:
                    launches.setdefault(k, {}).setdefault(grp[0], 0)
                    launches[k][grp[0]] += 1
    kernels = {}
    for k, a in agg.items():
        if "k_wf" not in k:
            continue
        n = max(launches.get(k, {}).get("FETCH_SIZE", 0), 1)
        fetch = a.get("FETCH_SIZE", 0.0) * 1024.0 * 2.0          # KiB -> bytes, x2: gfx950 wide-read correction
        write = a.get("WRITE_SIZE", 0.0) * 1024.0
        hit, miss = a.get("TCC_HIT_sum", 0.0), a.get("TCC_MISS_sum", 0.0)
        kernels[k] = {"launches": n, "fetch_bytes_per_launch": fetch / n, "write_bytes_per_launch": write / n,
                      "traffic_bytes_per_launch": (fetch + write) / n, "total_bytes": fetch + write,
                      "l2_hit_rate": hit / max(hit + miss, 1.0)}
    entry = {"workload": f"{scene} 1920x1080 {spp}spp n_gpus=1 accel={accel}", "options": opts, "calls": calls,
             "kernel_sources_sha256": sources_sha(),
             "method": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum in separate passes; FETCH_SIZE x 2 (gfx950), KiB -> bytes",
             "kernels": kernels}
    data = []
    if os.path.exists(out):
        data = json.load(open(out))
    data = [e for e in data if not (e["workload"] == entry["workload"] and e.get("options") == opts)] + [entry]
    json.dump(data, open(out, "w"), indent=1)
    for k, v in sorted(kernels.items()):
        print("%-28s launches %4d  fetch %.3f GB  write %.3f GB per launch  L2 hit %.3f" % (k, v["launches"], v["fetch_bytes_per_launch"] / 1e9, v["write_bytes_per_launch"] / 1e9, v["l2_hit_rate"]))


if __name__ == "__main__":
    main()
