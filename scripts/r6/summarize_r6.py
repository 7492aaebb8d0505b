#!/usr/bin/env python3
"""Round 6 version of scripts/r5/summarize_r5.py (the bench line of round 6: `roofline` is the vertex-id launch,
`roofline_pipeline` the ranks-out launch; cfg5).

Round 5 version of scripts/r4/summarize_r4.py (the bench line of round 5: workload string, value_vertex_ids).

Round 4 version of scripts/r3/summarize_r3.py (adds the wedge-slots kernel).

Round 3 version of scripts/summarize_r2.py.  Differences: the SGNS kernels' reads are counted
at HALF their bytes by FETCH_SIZE (calibrated in round 3 on the kernel's own access shape, one
wave reading a 512-byte row 8 bytes per lane: profiles/r3e_fetch_write_calibration_rows.txt --
256 B counted per 512-byte row read, 512 B per 512-byte row written), so their memory-side bytes
are 2 * FETCH_SIZE + WRITE_SIZE; the walk kernels' gathers stay as counted (64 B per gather,
profiles/r02_gather_fetch_calibration.txt).  The batched SGNS kernel gets its own entry.

Condense gpurun_out/<tag>/ (scripts/r4/profile_r4.sh: rocprofv3 --kernel-trace --stats plus
separate --pmc passes of bench.py) into profiles/<tag>_summary.json and
profiles/<tag>_kernel_stats.csv, and merge the measured bytes per launch into
profiles/pmc_traffic.json (read by bench.py's roofline).

    python scripts/summarize_r2.py r02_cfg4 cfg4

Counter handling (MI355X_MICROARCH.md "HBM"): FETCH_SIZE / WRITE_SIZE are in KB and count
requests on the memory side of L2 (Infinity-Cache hits included).  The guide's x2 correction
holds for wide coalesced reads (128-byte requests tallied at 64 B) and says other widths must
be calibrated: profiles/r02_gather_fetch_calibration.txt does that for THIS access pattern --
2^27 independent random 4-byte (and 16-byte) reads count 64.0 B each -- so the walk kernels'
gathers are taken as counted; for the SGNS kernel (512-byte rows read 8 B per lane) both
readings are reported and the uncorrected one is used (the corrected one would exceed what HBM
can deliver in the measured time).  Only full-size dispatches are averaged (setup launches
of the same kernel on 64 start vertices are dropped: values below half the maximum)."""
import csv
import glob
import json
import os
import sys

tag, config = sys.argv[1], sys.argv[2]
src = os.path.join("gpurun_out", tag)
os.makedirs("profiles", exist_ok=True)
pmc = json.load(open(os.path.join(src, "pmc_summary.json")))


def full(values):
    top = max(values)
    keep = [v for v in values if v >= 0.5 * top]
    return sum(keep) / len(keep), len(keep)


def counter(passname, kernel, cname):
    for key, ent in pmc.get(passname, {}).items():
        k, c = [x.strip() for x in key.split("|")]
        if kernel in k and c == cname:
            return full(ent["values"])
    return None, 0


bench = None
for line in open(os.path.join(src, "trace.log")):
    line = line.strip()
    if line.startswith("{") and '"metric"' in line:
        bench = json.loads(line)

kernels = {"walk_uniform_kernel<3, 1024>": "exact p=q=1, degree-ranked 4-byte table (ranks out): the headline",
           "walk_uniform_kernel<1, 256>": "exact p=q=1, 16-byte hop table (vertex ids out)",
           "walk_uniform_kernel<2, 256>": "exact p=q=1, 8-byte hop table (vertex ids out)",
           "walk_uniform_kernel<0, 256>": "exact p=q=1, CSR arrays",
           "walk_exact_wedge_kernel": "exact biased (all tables)",
           "walk_exact_wedge_slots_kernel": "exact biased (all tables, wedge slots)",
           "walk_exact_unit_lanes_kernel": "exact biased",
           "walk_exact_unit_kernel": "exact biased (wave per walker)",
           "walk_fast_kernel": "fast", "sgns_kernel": "sgns (gensim sampling)",
           "sgns_batched_kernel": "sgns, batched (negatives shared per centre position)"}
ROW_KERNELS = ("sgns_kernel", "sgns_batched_kernel")  # FETCH_SIZE counts half of their read bytes
out = {"tag": tag, "config": config, "kernels": {}}
stats_files = sorted(glob.glob(f"{src}/trace/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime,
                     reverse=True)  # the newest run of the tag
trace = {}
if stats_files:
    rows = list(csv.DictReader(open(stats_files[0])))
    with open(f"profiles/{tag}_kernel_stats.csv", "w") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys())
        w.writeheader()
        for r in rows[:14]:
            r["Name"] = r["Name"][:140]
            w.writerow(r)
    for r in rows:
        for k in kernels:
            if "::" + k + "(" in r["Name"] or "::" + k + "<" in r["Name"] or ("<" in k and "::" + k in r["Name"]):
                trace[k] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6,
                            "max_ms": float(r["MaxNs"]) / 1e6, "min_ms": float(r["MinNs"]) / 1e6}
for k, what in kernels.items():
    f, nf = counter("pmc_fetch", k, "FETCH_SIZE")
    if f is None:
        continue
    w, _ = counter("pmc_write", k, "WRITE_SIZE")
    hit, _ = counter("pmc_tcc", k, "TCC_HIT_sum")
    miss, _ = counter("pmc_tcc", k, "TCC_MISS_sum")
    ent = {"what": what, "full_size_dispatches": nf,
           "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w,
           "hbm_bytes_per_launch": ((2.0 if k in ROW_KERNELS else 1.0) * f + (w or 0.0)) * 1024.0,
           "hbm_bytes_per_launch_fetch_as_counted": (f + (w or 0.0)) * 1024.0,
           "fetch_correction": 2.0 if k in ROW_KERNELS else 1.0,
           "tcc_hit_rate": None if hit is None else hit / max(hit + miss, 1.0)}
    for c in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
              "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_INSTS_VALU"):
        v, _ = counter("pmc_sq", k, c)
        if v is not None:
            ent[c] = v
    if "SQ_WAVE_CYCLES" in ent:
        wc = ent["SQ_WAVE_CYCLES"]
        ent["share_of_wave_cycles"] = {n: ent[c] / wc for n, c in (
            ("waiting_any", "SQ_WAIT_ANY"), ("waiting_inst_issue", "SQ_WAIT_INST_ANY"),
            ("issuing_any", "SQ_ACTIVE_INST_ANY"), ("issuing_valu", "SQ_ACTIVE_INST_VALU"),
            ("issuing_scalar", "SQ_ACTIVE_INST_SCA")) if c in ent}
    if k in trace:
        ent["kernel_trace"] = trace[k]
        ms = trace[k]["max_ms"] if trace[k]["min_ms"] < 0.5 * trace[k]["max_ms"] else trace[k]["avg_ms"]
        ent["TBps_memory_side"] = ent["hbm_bytes_per_launch"] / (ms * 1e-3) / 1e12
        ent["frac_of_8TBps"] = ent["TBps_memory_side"] / 8.0
    out["kernels"][k] = ent
if bench:
    out["bench_line_of_the_trace_run"] = bench
json.dump(out, open(f"profiles/{tag}_summary.json", "w"), indent=1)

# profiles/pmc_traffic.json: key = config:kernel:p:q:batch (bench.py pmc_traffic)
tpath = "profiles/pmc_traffic.json"
table = json.load(open(tpath)) if os.path.exists(tpath) else {}
if bench:
    def put(kernel, p, q, batch, hops=False, wedges=False):
        if kernel == "walk_uniform_kernel":  # one entry per instance of the template
            inst = {"4-byte degree-ranked": "<3, 1024>", "8-byte": "<2, 256>"}.get(hops, "<1, 256>" if hops else "<0, 256>")
            e = out["kernels"].get(kernel + inst)
        else:
            e = out["kernels"].get(kernel)
        if e:
            tab = ":ranked" if hops == "4-byte degree-ranked" else ":hop8" if hops == "8-byte" else (":hops" if hops else "")
            table[f"{config}:{kernel}{tab}{':wedges' if wedges else ''}:p{p}:q{q}:batch{batch}"] = {
                "hbm_bytes_per_launch": e["hbm_bytes_per_launch"], "source": f"profiles/{tag}_summary.json",
                "FETCH_SIZE_KB": e["FETCH_SIZE_KB"], "WRITE_SIZE_KB": e["WRITE_SIZE_KB"],
                "tcc_hit_rate": e["tcc_hit_rate"]}
    head = bench["roofline"]["kernel"]
    hb = int(bench["config"]["start_vertices_per_step_per_gpu"])
    pq = bench["config"]["workload"].split("p=")[1].split(",")[0].split()
    put(head, float(pq[0]), float(pq[1].replace("q=", "")), hb, bench["roofline"].get("hop_table", False))
    if "roofline_pipeline" in bench:
        put(bench["roofline_pipeline"]["kernel"], float(pq[0]), float(pq[1].replace("q=", "")), hb,
            bench["roofline_pipeline"].get("hop_table", False))
    if "biased" in bench:
        b = bench["biased"]
        put(b["roofline"]["kernel"], b["p"], b["q"], b["start_vertices_per_step"],
            b["roofline"].get("hop_table", False), b["roofline"].get("wedge_table", False))
    if "fast_mode" in bench:
        b = bench["fast_mode"]
        put("walk_fast_kernel", b["p"], b["q"], b["start_vertices_per_step"],
            b["roofline"].get("hop_table", False))
    if "sgns" in bench:
        put("sgns_kernel", 0, 0, bench["sgns"]["config"]["rows_per_step"])
        if "batched" in bench["sgns"]:
            put("sgns_batched_kernel", 0, 0, bench["sgns"]["config"]["rows_per_step"])
json.dump(table, open(tpath, "w"), indent=1)
print(json.dumps({k: {a: b for a, b in v.items() if not a.startswith("SQ_")} for k, v in out["kernels"].items()}, indent=1))
