"""Summarise rocprofv3 output directories (kernel stats, PMC passes) into small text / csv files.
usage: collect_profile.py OUT_DIR KERNEL_REGEX stats=DIR fetch=DIR write=DIR [mfma=DIR sq1=DIR sq2=DIR]
Writes OUT_DIR/kernel_stats.csv (the --stats table), OUT_DIR/launches.csv (per-dispatch resources of the matching kernels:
VGPRs, LDS, workgroup / grid size -> occupancy), OUT_DIR/pmc_<pass>.txt (counter sums over the matching kernels) and prints
the derived figures (HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE in KB per the gfx950 rule of MI355X_MICROARCH.md; MFMA pipe
busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs); wave-time split from SQ_WAIT_* / SQ_ACTIVE_*)."""
import csv, glob, json, os, re, shutil, sys

out, pat = sys.argv[1], re.compile(sys.argv[2])
os.makedirs(out, exist_ok=True)
dirs = {k: v for k, v in (a.split("=", 1) for a in sys.argv[3:]) if os.path.isdir(v)}


def counters(d):
    agg, n = {}, {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if pat.search(r["Kernel_Name"]):
                agg[r["Counter_Name"]] = agg.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                n[r["Kernel_Name"].split("(")[0][:60]] = 1
    return agg, sorted(n)


if "stats" in dirs:
    for f in glob.glob(os.path.join(dirs["stats"], "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(out, "kernel_stats.csv"))
    rows = []
    for f in glob.glob(os.path.join(dirs["stats"], "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if pat.search(r["Kernel_Name"]):
                rows.append(r)
    if rows:
        keep = [k for k in ("Kernel_Name", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size",
                            "Workgroup_Size", "Grid_Size", "Start_Timestamp", "End_Timestamp") if k in rows[0]]
        with open(os.path.join(out, "launches.csv"), "w", newline="") as fh:
            w = csv.writer(fh)
            # (the trace reports the architected half of the unified VGPR file and no dynamic LDS: occupancy is stated in
            #  DESIGN.md from the compiler's resource report and the launch's dynamic LDS size)
            w.writerow(keep + ["duration_ms"])
            for r in rows:
                dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
                w.writerow([r[k][:70] if k == "Kernel_Name" else r[k] for k in keep] + [f"{dur:.3f}"])
res = {}
for key in ("fetch", "write", "mfma", "sq1", "sq2"):
    if key in dirs:
        c, names = counters(dirs[key])
        with open(os.path.join(out, f"pmc_{key}.txt"), "w") as fh:
            for n_, v in sorted(c.items()):
                fh.write(f"{n_}: {v:.6g}\n")
            fh.write(f"(rocprofv3 --kernel-trace --pmc ..., counter sums over ALL launches of {names} in ONE run of the command)\n")
        res.update(c)
d = {}
if "FETCH_SIZE" in res and "WRITE_SIZE" in res:
    d.update(fetch_size_kb=res["FETCH_SIZE"], write_size_kb=res["WRITE_SIZE"],
             hbm_bytes=(2.0 * res["FETCH_SIZE"] + res["WRITE_SIZE"]) * 1024.0)
if "SQ_VALU_MFMA_BUSY_CYCLES" in res and res.get("GRBM_GUI_ACTIVE"):
    d["mfma_pipe_busy"] = res["SQ_VALU_MFMA_BUSY_CYCLES"] / (res["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
if res.get("SQ_WAVE_CYCLES"):
    wc = res["SQ_WAVE_CYCLES"]
    d["wave_time_split"] = {"parked_waitcnt_or_barrier": res.get("SQ_WAIT_ANY", 0) / wc, "issue_stalled": res.get("SQ_WAIT_INST_ANY", 0) / wc,
                            "issuing": res.get("SQ_ACTIVE_INST_ANY", 0) / wc}
if res.get("SQ_LDS_IDX_ACTIVE"):
    d["lds_bank_conflict_frac"] = res.get("SQ_LDS_BANK_CONFLICT", 0) / res["SQ_LDS_IDX_ACTIVE"]
for k in ("SQ_INSTS_VALU_MFMA_F32", "SQ_INSTS_VALU_MFMA_F64", "SQ_INSTS_VALU"):
    if k in res:
        d[k] = res[k]
print(json.dumps(d, indent=1))
