#!/usr/bin/env python3
"""Sum rocprofv3 PMC passes per kernel and write the per-launch HBM traffic bench.py quotes.

    python profiles/aggregate_pmc.py <tag> <workload> <dir with *_counter_collection.csv files...>

Each directory is one `rocprofv3 --pmc <counters> --output-format csv` pass of
`python3 bench.py --workload <workload> --steps 1 --warmup 1 --no-cpu-baseline` (counters must
be collected in separate passes from the kernel trace; FETCH_SIZE and WRITE_SIZE do not fit one
pass).  Output: profiles/<tag>_<workload>_pmc_by_kernel.json (raw sums per kernel and half-step)
and the <workload> entry of profiles/traffic.json.

HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KB: on gfx950 FETCH_SIZE counts a 128-byte
request of a 16-byte-per-lane load as 64 bytes (MI355X_MICROARCH.md, HBM); every gather of these
kernels is 16 bytes per lane.  Infinity-Cache hits are included in FETCH_SIZE, so this is
memory-side traffic of the L2, an upper bound of the HBM bytes.
"""
import csv
import glob
import json
import os
import sys

# kernel name fragment -> bench.py's entry (the workgroup-per-row kernels of k > 128, als_wg_*, fill the same roles)
BENCH_NAME = (("gram_solve", "als_gram_solve_kernel"), ("dual_solve", "als_dual_solve_kernel"),
              ("gram_slab", "als_gram_slab_kernel"), ("reduce_solve", "als_reduce_solve_kernel"),
              # round 4: rows of many slabs are folded in groups before the reduce; its traffic belongs to the reduce entry
              ("slab_fold", "als_reduce_solve_kernel"),
              # k > 240: whole rows go Gramian -> slab -> few-wave solve in BATCHES (als_pair_kernels.hip.h); together they
              # are bench.py's als_gram_solve_kernel entry, one "launch" = all batches of a half-step
              ("gram_rowslab", "als_gram_solve_kernel"), ("slab_solve2", "als_gram_solve_kernel"))
BATCHED = ("gram_rowslab", "slab_solve2")
ITERATIONS = 2  # collect.sh: --steps 1 --warmup 1


def main():
    tag, workload, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    rows = {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                rows.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"]})[r["Counter_Name"]] = \
                    rows.get(int(r["Dispatch_Id"]), {}).get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    # half-step of each dispatch.  Dispatch order within a half-step: [slab] [dual classes] [row
    # kernel] [reduce] with the dual classes on side streams (they are enqueued before the row
    # kernel), or [slab] [row kernel] [dual classes] [reduce] in stream order; k > 128:
    # ([gram_big] [solve_big])* [dual classes].  A new half-step begins after a reduce kernel, at
    # a slab kernel that follows anything but slab / big kernels, and at a kernel name that the
    # current half-step already holds.
    per = {}
    order = [d for d in sorted(rows) if "ycnr::als_" in rows[d]["name"] and "rmse" not in rows[d]["name"]]
    sides = {}
    half, seen, prev = 1, set(), ""
    for did in order:
        n = rows[did]["name"].split("(")[0]
        if any(b in n for b in BATCHED):  # the batches of a half-step alternate two kernels: no half-step boundary among them
            n = "batched_rows"
        slab = "gram_slab" in n
        new_half = False
        if seen:
            if "reduce_solve" in prev:
                new_half = True
            elif slab and "gram_slab" not in prev and "split_planes" not in prev:  # (the planes kernel opens a half-step in front of its slab kernel)
                new_half = True
            elif n in seen and not slab and n != "batched_rows":
                new_half = True
        if new_half:
            half += 1
            seen = set()
        seen.add(n)
        prev = n
        sides[did] = "byUser" if half % 2 == 1 else "byItem"
    for did in order:
        r = rows[did]
        short = r["name"].split("(")[0].replace("void ", "")
        if "ycnr::als_" not in short or "rmse" in short:
            continue
        key = (short, sides[did])
        d = per.setdefault(key, {"kernel": short, "half_step": sides[did], "dispatches": 0})
        d["dispatches"] += 1
        for c, v in r.items():
            if c != "name":
                d[c] = d.get(c, 0.0) + v
    out = sorted(per.values(), key=lambda d: (d["half_step"], d["kernel"]))
    here = os.path.dirname(os.path.abspath(__file__))
    json.dump(out, open(os.path.join(here, f"{tag}_{workload}_pmc_by_kernel.json"), "w"), indent=1)
    # per-launch traffic under bench.py's kernel names (all dual classes of a half-step are one bench entry)
    traffic = {}
    for d in out:
        bn = next((b for a, b in BENCH_NAME if a in d["kernel"]), None)
        if bn is None or "FETCH_SIZE" not in d:
            continue
        t = traffic.setdefault(f"{bn}[{d['half_step']}]", {"kb": 0.0, "launches": 0})
        t["kb"] += 2.0 * d["FETCH_SIZE"] + d.get("WRITE_SIZE", 0.0)
        t["launches"] = max(t["launches"], ITERATIONS if any(b in d["kernel"] for b in BATCHED) else d["dispatches"])
    # the whole-row kernels of a half-step as one group (bench.py's entry when the dual kernels run
    # on side streams next to the row kernel)
    for side in ("byUser", "byItem"):
        a, b = traffic.get(f"als_gram_solve_kernel[{side}]"), traffic.get(f"als_dual_solve_kernel[{side}]")
        if a and b:
            traffic[f"als_gram_solve_kernel+als_dual_solve_kernel[{side}]"] = {
                "kb": a["kb"] / a["launches"] + b["kb"] / b["launches"], "launches": 1}
        # round 4: with a side stream per dual class the chunk kernel of the step's stream shares the group's interval too
        g, c = traffic.get(f"als_gram_solve_kernel+als_dual_solve_kernel[{side}]"), traffic.get(f"als_gram_slab_kernel[{side}]")
        if g and c:
            traffic[f"als_gram_solve_kernel+als_dual_solve_kernel+als_gram_slab_kernel[{side}]"] = {
                "kb": g["kb"] / g["launches"] + c["kb"] / c["launches"], "launches": 1}
    tpath = os.path.join(here, "traffic.json")
    tj = json.load(open(tpath)) if os.path.exists(tpath) else {"workloads": {}}
    tj.setdefault("sources", {})[workload] = (f"profiles/{tag}_{workload}_pmc_by_kernel.json: (2 x FETCH_SIZE + WRITE_SIZE) KB per launch, "
                                              "rocprofv3 --pmc, own passes")
    try:  # which tree the counters were measured on (bench.py prints it beside roofline.traffic)
        import subprocess
        tj.setdefault("commits", {})[workload] = subprocess.check_output(["git", "-C", here, "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:  # noqa: BLE001
        pass
    tj["workloads"][workload] = {k: {"hbm_bytes_per_launch": round(v["kb"] * 1024.0 / v["launches"])} for k, v in traffic.items()}
    json.dump(tj, open(tpath, "w"), indent=1)
    for k, v in tj["workloads"][workload].items():
        print(k, v)


if __name__ == "__main__":
    main()
