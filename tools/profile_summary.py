#!/usr/bin/env python3
"""Turns raw rocprofv3 output (gpurun_out/...) into the summaries committed under profiles/.

    # on the GPU box (three separate runs; never --pmc together with a trace):
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o run -- python3 bench.py --steps 2 --warmup 1 \
        --no-cpu-baseline --no-stream --no-timing --workers 1
    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o run -- python3 bench.py --steps 1 --warmup 0 \
        --no-cpu-baseline --no-stream --no-timing --workers 1
    rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write -o run -- python3 bench.py --steps 1 --warmup 0 \
        --no-cpu-baseline --no-stream --no-timing --workers 1
    # here:
    python tools/profile_summary.py stats gpurun_out/prof profiles/rNN_cfg3_kernel_stats.csv "<command line>"
    python tools/profile_summary.py traffic gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/rNN_cfg3_pmc_hbm_traffic.csv

`traffic` also rewrites profiles/traffic.json (HBM bytes per launch of the kernels bench.py reports).  FETCH_SIZE and
WRITE_SIZE are in KiB; FETCH_SIZE under-reports 128-byte requests by 2x on gfx950 (MI355X_MICROARCH.md), which the
1.7 GB device-to-device splat copy of every bench pass confirms (it reads as 0.49x its size), so reads are doubled.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# kernels of the bench's per-stage table -> substring of the demangled kernel name
PMC_PASSES = 2      # tools/profile_run.sh's counter passes run `--steps 1 --warmup 0`: the counting pass and one timed pass
PMC_PASSES_OF = {"processCorners": 1}       # ... and the counting pass runs the instrumented processCorners (<0, true>), not the tracked one
TRACKED = {
    "processCorners": "processCornersMatrixKernel<0, false>",
    "latticeTriangles": "latticeTriangles",      # by rows (noise cloud) or by cells (surface-like data)
    "latticeVertices": "latticeVerticesKernel",
    "latticeMask": "latticeMaskWordKernel",
    "sortScatter": "sortScatterKernel",
    "sortHist": "sortHistKernel",
    "cellCode": "cellCodeKernel",
    "writeEntries": "entryScatterKernel",
    "writeSplatIds": "SplatIdsOut",             # rounds 1-3 and the deep-tree route; gone from the default route in round 4
}


def find(directory, suffix):
    hits = glob.glob(os.path.join(directory, "**", "*" + suffix), recursive=True)
    if not hits:
        raise SystemExit("no *%s under %s" % (suffix, directory))
    return sorted(hits)[-1]


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def database(directory):
    """rocprofv3's default output here is a rocpd SQLite file; --output-format csv gives the csv files instead"""
    hits = glob.glob(os.path.join(directory, "**", "*_results.db"), recursive=True)
    if not hits:
        return None
    import sqlite3
    return sqlite3.connect(sorted(hits)[-1])


def stats_rows(directory):
    db = database(directory)
    if db is None:
        return list(csv.DictReader(open(find(directory, "kernel_stats.csv"))))
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                      "group by name order by sum(duration) desc").fetchall()
    whole = sum(r[2] for r in rows)
    return [dict(Name=r[0], Calls=r[1], TotalDurationNs=r[2], AverageNs="%.1f" % r[3], Percentage="%.4f" % (100.0 * r[2] / whole),
                 MinNs=r[4], MaxNs=r[5]) for r in rows]


def stats(directory, out, command):
    rows = stats_rows(directory)
    with open(out, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- %s\n" % command)
        f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
        for r in rows:
            f.write('"%s",%s,%s,%s,%s,%s,%s\n' % (short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"],
                                                 r["Percentage"], r["MinNs"], r["MaxNs"]))
    print("wrote", out, len(rows), "kernels")


def counters(directory, counter):
    total = defaultdict(float)
    calls = defaultdict(int)
    db = database(directory)
    if db is not None:
        for name, value in db.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
            total[short(name)] += float(value)
            calls[short(name)] += 1
        return total, calls
    for r in csv.DictReader(open(find(directory, "counter_collection.csv"))):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        total[k] += float(r["Counter_Value"])
        calls[k] += 1
    return total, calls


def traffic(fetch_dir, write_dir, out):
    fetch, calls = counters(fetch_dir, "FETCH_SIZE")
    write, wcalls = counters(write_dir, "WRITE_SIZE")
    names = sorted(set(fetch) | set(write), key=lambda k: -(2 * fetch.get(k, 0) + write.get(k, 0)))
    with open(out, "w") as f:
        f.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes), bench.py --steps 1 --warmup 0 --workers 1\n")
        f.write("# counters are in KiB; fetch bytes are 2 x FETCH_SIZE x 1024 (gfx950 correction, see tools/profile_summary.py)\n")
        f.write("kernel,calls,FETCH_SIZE_KB_total,WRITE_SIZE_KB_total,fetch_bytes_per_launch_x2corrected,write_bytes_per_launch\n")
        for k in names:
            n = max(calls.get(k, 0), wcalls.get(k, 0), 1)
            f.write('"%s",%d,%.1f,%.1f,%.0f,%.0f\n' % (k, n, fetch.get(k, 0), write.get(k, 0),
                                                       2 * 1024 * fetch.get(k, 0) / n, 1024 * write.get(k, 0) / n))
    per_launch, per_pass = {}, {}
    for label, needle in TRACKED.items():
        fk = [k for k in names if needle in k]
        if not fk:
            continue
        n = sum(max(calls.get(k, 0), wcalls.get(k, 0)) for k in fk)
        moved = 2 * 1024 * sum(fetch.get(k, 0) for k in fk) + 1024 * sum(write.get(k, 0) for k in fk)
        per_launch[label] = int(moved / max(n, 1))
        per_pass[label] = int(moved / PMC_PASSES_OF.get(label, PMC_PASSES))
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    doc = json.load(open(tpath)) if os.path.exists(tpath) else {}
    doc["cfg3/uniform"] = per_launch
    doc["cfg3/uniform/per_pass"] = per_pass
    doc["_note_per_pass"] = ("bytes of all the kernel's launches in one pass over the workload's buckets (a launch covers the "
                             "buckets of a batch -- two for processCorners and the marching kernels, four for the octree's, by "
                             "default): the profiled command runs %d passes" % PMC_PASSES)
    doc["_note"] = ("HBM bytes per launch = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 fetch correction), from "
                    + os.path.relpath(out, ROOT))
    json.dump(doc, open(tpath, "w"), indent=1)
    print("wrote", out, "and", tpath, per_launch)


def traffic_resolved(rd_dir, wr_dir, out):
    """HBM traffic per launch from the request-size-resolved TCC counters (round 3).  rocprofv3's FETCH_SIZE
    is (BUBBLE*128 + (RDREQ - BUBBLE - RDREQ_32B)*64 + RDREQ_32B*32) / 1024: on gfx950 the 128-byte read requests are in
    TCC_EA0_RDREQ_128B, not in TCC_BUBBLE, so FETCH_SIZE tallies them at 64 bytes -- exactly half for a kernel whose reads
    are all 128-byte requests, exact for one whose reads are 32- or 64-byte requests.  Here every size is counted at its own
    width, per kernel; the csv also gives what FETCH_SIZE would have said and the ratio (the kernel's own correction)."""
    cols = {}
    calls = {}
    for c in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"):
        cols[c], calls = counters(rd_dir, c)
    wcalls = {}
    for c in ("TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"):
        cols[c], wcalls = counters(wr_dir, c)
    names = sorted(set(cols["TCC_EA0_RDREQ_sum"]) | set(cols["TCC_EA0_WRREQ_sum"]),
                   key=lambda k: -(cols["TCC_EA0_RDREQ_sum"].get(k, 0) + cols["TCC_EA0_WRREQ_sum"].get(k, 0)))

    def row(k):
        r, r32, r64, r128 = (cols[c].get(k, 0.0) for c in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum",
                                                           "TCC_EA0_RDREQ_128B_sum"))
        w, w64 = cols["TCC_EA0_WRREQ_sum"].get(k, 0.0), cols["TCC_EA0_WRREQ_64B_sum"].get(k, 0.0)
        fetch = 32 * r32 + 64 * r64 + 128 * r128
        formula = 32 * r32 + 64 * (r - r32)
        write = 64 * w64 + 32 * (w - w64)
        return r, r32, r64, r128, w, w64, fetch, formula, write
    with open(out, "w") as f:
        f.write("# rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum / --pmc "
                "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum (two separate passes, no tracing), bench.py --headline-only --no-timing "
                "--workers 1 --steps 1 --warmup 0\n")
        f.write("# fetch bytes = 32*R32 + 64*R64 + 128*R128; FETCH_SIZE's formula counts the 128-byte requests at 64 (its TCC_BUBBLE "
                "term is empty on gfx950): factor = fetch / formula is the kernel's own correction, between 1 and 2\n")
        f.write("kernel,calls,RDREQ,RDREQ_32B,RDREQ_64B,RDREQ_128B,sizes_add_up,fetch_bytes_per_launch,FETCH_SIZE_formula_bytes_per_launch,"
                "factor,WRREQ,WRREQ_64B,write_bytes_per_launch\n")
        for k in names:
            r, r32, r64, r128, w, w64, fetch, formula, write = row(k)
            n = max(calls.get(k, 0), wcalls.get(k, 0), 1)
            f.write('"%s",%d,%.0f,%.0f,%.0f,%.0f,%s,%.0f,%.0f,%.3f,%.0f,%.0f,%.0f\n'
                    % (k, n, r, r32, r64, r128, "yes" if abs(r - r32 - r64 - r128) <= 1e-6 * max(r, 1) else "no", fetch / n,
                       formula / n, fetch / formula if formula else 0.0, w, w64, write / n))
    per_launch, per_pass, factors = {}, {}, {}
    for label, needle in TRACKED.items():
        fk = [k for k in names if needle in k]
        if not fk:
            continue
        n = sum(max(calls.get(k, 0), wcalls.get(k, 0)) for k in fk)
        rows = [row(k) for k in fk]
        per_launch[label] = int((sum(x[6] for x in rows) + sum(x[8] for x in rows)) / max(n, 1))
        per_pass[label] = int((sum(x[6] for x in rows) + sum(x[8] for x in rows)) / PMC_PASSES_OF.get(label, PMC_PASSES))
        factors[label] = round(sum(x[6] for x in rows) / max(sum(x[7] for x in rows), 1), 3)
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    doc = json.load(open(tpath)) if os.path.exists(tpath) else {}
    doc["cfg3/uniform"] = per_launch
    doc["cfg3/uniform/per_pass"] = per_pass
    doc["_fetch_correction_per_kernel"] = factors
    doc["_note_per_pass"] = ("bytes of all the kernel's launches in one pass over the workload's buckets (a launch covers the "
                             "buckets of a batch -- two for processCorners and the marching kernels, four for the octree's, by "
                             "default): the profiled command runs %d passes" % PMC_PASSES)
    doc["_note"] = ("HBM bytes per launch = 32*RDREQ_32B + 64*RDREQ_64B + 128*RDREQ_128B + 64*WRREQ_64B + 32*(WRREQ - WRREQ_64B), every "
                    "request at its own width, from " + os.path.relpath(out, ROOT) + ".  _fetch_correction_per_kernel is what a "
                    "FETCH_SIZE reading of that kernel would have to be multiplied by (2.0 = all reads are 128-byte requests, the "
                    "guide's streaming case; 1.0 = none are): rounds 1-2 doubled every kernel.")
    json.dump(doc, open(tpath, "w"), indent=1)
    print("wrote", out, "and", tpath, per_launch, factors)


def sq(dirs, label, out):
    """Sums of every collected counter over the launches of processCorners; appended to `out` as label,counter,value."""
    rows = {}
    launches = 0
    # MLSGPU_SQ_KERNELS=a,b,c: other kernels (substrings of the name), one block of rows per kernel
    spec = os.environ.get("MLSGPU_SQ_KERNELS", "processCorners")
    kernels = [k for k in spec.split(";" if ";" in spec else ",") if k]      # "a;" = the one substring "a" (it may hold commas)
    for d in dirs:
        db = database(d)
        if db is not None:
            it = db.execute("select kernel_name, counter_name, value from counters_collection")
        else:
            it = ((r["Kernel_Name"], r["Counter_Name"], r["Counter_Value"]) for r in csv.DictReader(open(find(d, "counter_collection.csv"))))
        seen = defaultdict(int)
        for name, counter, value in it:
            hit = [k for k in kernels if k in name]
            if not hit:
                continue
            key = (hit[0], counter) if len(kernels) > 1 else counter
            rows[key] = rows.get(key, 0.0) + float(value)
            seen[key] += 1
        if seen:
            launches = max(launches, max(seen.values()))
    new = not os.path.exists(out)
    with open(out, "a") as f:
        if new:
            f.write("# rocprofv3 --pmc (passes of <= 8 SQ counters each, no tracing) over bench.py --headline-only --no-timing "
                    "--workers 1 --steps 1 --warmup 0 [workload given in the label; tools/sq_counters*.sh]\n")
            f.write("# sums over the launches of the named kernel(s) in the run (`launches`); SQ_WAVE_CYCLES / SQ_WAIT_* / "
                    "SQ_ACTIVE_INST_* are in quad-cycles (MI355X_MICROARCH.md)\n")
            f.write("variant,launches,counter,value\n")
        for c in sorted(rows):
            if isinstance(c, tuple):
                f.write('"%s: %s",%d,%s,%.0f\n' % (label, c[0], launches, c[1], rows[c]))
            else:
                f.write('"%s",%d,%s,%.0f\n' % (label, launches, c, rows[c]))
    print("appended", len(rows), "counters for", label, "to", out)
    # MLSGPU_SQ_JSON=<file>:<key>: the figures bench.py puts into its line (roofline.valu_issue_frac, ...), for ONE kernel
    spec = os.environ.get("MLSGPU_SQ_JSON")
    if spec and all(not isinstance(c, tuple) for c in rows) and "GRBM_GUI_ACTIVE" in rows and "SQ_INSTS_VALU" in rows:
        path, key = spec.rsplit(":", 1)
        try:
            doc = json.load(open(path))
        except Exception:
            doc = {}
        cycles = rows["GRBM_GUI_ACTIVE"] / 8.0               # the counter is summed over the 8 XCDs
        simds, cus = 1024.0, 256.0
        doc[key] = {
            "label": label, "launches": launches, "kernel_cycles": int(cycles),
            "insts_valu": int(rows["SQ_INSTS_VALU"]), "insts_lds": int(rows.get("SQ_INSTS_LDS", 0)),
            "insts_salu": int(rows.get("SQ_INSTS_SALU", 0)), "insts_mfma": int(rows.get("SQ_INSTS_MFMA", 0)),
            # a wave's vector instruction occupies its SIMD-32 for 2 cycles (MI355X_MICROARCH.md, wave scheduling); the
            # shortest interval a SIMD sustained with 8 resident waves was 2.63 (profiles/r03_valu_lds_issue_microbench.txt)
            "valu_issue_frac": round(rows["SQ_INSTS_VALU"] * 2.0 / (cycles * simds), 4),
            "valu_issue_frac_of_measured_rate": round(rows["SQ_INSTS_VALU"] * 2.63 / (cycles * simds), 4),
            "mfma_busy_frac": round(rows.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cycles * simds), 4),
            "lds_busy_frac": round(rows.get("SQ_LDS_IDX_ACTIVE", 0) / (cycles * cus), 4),
            "lds_bank_conflict_frac": round(rows.get("SQ_LDS_BANK_CONFLICT", 0) / max(rows.get("SQ_LDS_IDX_ACTIVE", 0), 1), 4),
            "active_lane_frac": round(rows.get("SQ_THREAD_CYCLES_VALU", 0) / (64.0 * max(rows["SQ_INSTS_VALU"], 1)), 4),
            "wave_wait_frac": round(rows.get("SQ_WAIT_ANY", 0) / max(rows.get("SQ_WAVE_CYCLES", 0), 1), 4),
            "source": os.path.relpath(out, ROOT),
        }
        json.dump(doc, open(path, "w"), indent=1, sort_keys=True)
        print("wrote", key, "to", path)


if __name__ == "__main__":
    if len(sys.argv) >= 5 and sys.argv[1] == "sq":
        sq(sys.argv[4:], sys.argv[2], sys.argv[3])
    elif len(sys.argv) >= 5 and sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3], sys.argv[4])
    elif len(sys.argv) == 5 and sys.argv[1] == "traffic_resolved":
        traffic_resolved(sys.argv[2], sys.argv[3], sys.argv[4])
    elif len(sys.argv) == 5 and sys.argv[1] == "traffic":
        traffic(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        raise SystemExit(__doc__)
