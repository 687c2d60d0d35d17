"""Collect rocprofv3 PMC counters of one kernel in separate passes (<= 8 SQ counters per pass; FETCH_SIZE / WRITE_SIZE alone)
and write a JSON summary.   python tools/pmc_collect.py OUT.json KERNEL_SUBSTR[,KERNEL_SUBSTR2=OUT2.json,...] -- python3 tools/prog.py
(run from the repo root ON THE GPU BOX; the profiled program goes after `--` unwrapped)"""
import csv, glob, json, os, subprocess, sys, tempfile

PASSES = [
    ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAVES", "SQ_INSTS_MFMA"],
    ["GRBM_GUI_ACTIVE", "SQ_INSTS_LDS", "SQ_INSTS_VALU", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM"],
    ["FETCH_SIZE"],
    ["WRITE_SIZE"],
    ["TA_BUSY_avr", "TCC_HIT_sum", "TCC_MISS_sum"],
]


def main():
    out, kerns = sys.argv[1], sys.argv[2].split(",")
    cmd = sys.argv[sys.argv.index("--") + 1:]
    dirs = []
    for ctrs in PASSES:
        d = tempfile.mkdtemp(prefix="pmc_", dir="/tmp")
        env = dict(os.environ, TMPDIR="/tmp")
        # (a pass is bounded: PMC_PASS_TIMEOUT seconds, default 600 -- a pass that does not finish is skipped and named, the others still count)
        try:
            r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc"] + ctrs + ["-d", d, "--output-format", "csv", "--"] + cmd,
                               env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=float(os.environ.get("PMC_PASS_TIMEOUT", "600")))
        except subprocess.TimeoutExpired:
            print("pass did not finish:", ctrs, flush=True)
            continue
        if r.returncode != 0:
            print("pass failed:", ctrs, r.stdout[-2000:])
            continue
        dirs.append(d)
    for i, k in enumerate(kerns):
        kern, _, o = k.partition("=")
        summarise(dirs, kern, o or out, cmd)


def summarise(dirs, kern, out, cmd):
    res, dur = {}, []
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            acc = {}
            for row in csv.DictReader(open(f)):
                if kern not in row.get("Kernel_Name", ""):
                    continue
                acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
            for k, v in acc.items():
                # first launch dropped (cold caches)
                v = v[1:] if len(v) > 1 else v
                res[k] = sum(v) / len(v)
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if kern in row.get("Kernel_Name", ""):
                    dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    summary = {"kernel": kern, "command": " ".join(cmd), "counters": res,
               "avg_us_per_launch_profiled": sum(dur) / max(1, len(dur)), "launches_seen": len(dur)}
    c = res
    if "GRBM_GUI_ACTIVE" in c and dur:
        wall_cycles = c["GRBM_GUI_ACTIVE"] / 8.0
        summary["derived"] = {"kernel_wall_cycles": wall_cycles, "effective_clock_GHz": wall_cycles / (sum(dur) / len(dur)) / 1e3}
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            summary["derived"]["mfma_busy_fraction_of_wall"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / wall_cycles
        if "SQ_WAVE_CYCLES" in c:
            wc = c["SQ_WAVE_CYCLES"]
            summary["derived"].update(wave_issue_stall_fraction=c.get("SQ_WAIT_INST_ANY", 0) / wc, wave_parked_fraction=c.get("SQ_WAIT_ANY", 0) / wc,
                                      wave_active_fraction=c.get("SQ_ACTIVE_INST_ANY", 0) / wc)
        if "SQ_LDS_IDX_ACTIVE" in c:
            summary["derived"]["lds_busy_fraction"] = c["SQ_LDS_IDX_ACTIVE"] / 256.0 / wall_cycles
    if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
        # MI355X_MICROARCH.md: FETCH_SIZE (KB) reports half the bytes of wide streaming reads on gfx950 -> x2; WRITE_SIZE exact
        summary["hbm_bytes_per_launch"] = c.get("FETCH_SIZE", 0) * 1024 * 2 + c.get("WRITE_SIZE", 0) * 1024
        summary["traffic_note"] = "FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE, KB -> bytes, separate passes"
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    json.dump(summary, open(out, "w"), indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
