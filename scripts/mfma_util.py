"""MFMA utilisation of the wide kernels from hardware counters (north_star: "choices evidenced by rocprof HBM GB/s and MFMA utilisation vs.
chip peak"), parity mode and plain-bf16 mode.  For every config runs bench.py (eager tapes) under
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace
(counters in their own pass: kernel-trace only) and, per wide kernel, prints: launches, mean duration (trace), MFMA-busy cycles per launch
(summed over the chip's 1024 SIMDs; = 32 cycles per v_mfma_f32_32x32x16_bf16, 16 per 16x16x32), and the busy fraction of the matrix pipes
= busy / (duration x 2.4 GHz x 1024) -- against the UN-throttled clock, i.e. the fraction of the chip's peak MFMA issue the kernel used.
    python scripts/mfma_util.py [out_dir]"""
import collections, csv, glob, os, re, subprocess, sys, shutil
out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r06_mfma"
os.makedirs(out, exist_ok=True)
exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
Q = ["--steps", "6", "--warmup", "2", "--no-graph", "--no-cpu-baseline", "--sweep-trials-per-gpu", "0", "--no-other", "--repeats", "0", "--no-pmc"]
WIDE = ("fx_dw_adam_fwd_kernel", "fx_gemm_bf16x3_kernel", "fx_fwd_bf16x3_reg_kernel")
print(f"{'config':6s} {'mode':7s} {'kernel':58s} {'n':>4s} {'us':>8s} {'MFMA busy Mcyc':>15s} {'pipe busy':>10s}")
for cfg in ("cfg2", "cfg3", "cfg4"):
    for prec in ("bf16x3", "bf16"):
        d = os.path.join(out, f"{cfg}_{prec}")
        shutil.rmtree(d, ignore_errors=True)
        env = dict(os.environ, TMPDIR="/tmp")
        r = subprocess.run([exe, "--pmc", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "--kernel-trace", "--output-format", "csv", "-d", d, "--",
                            sys.executable, os.path.abspath("bench.py"), "--config", cfg, "--precision", prec] + Q, env=env, capture_output=True, text=True)
        cc = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))
        kt = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))
        if not cc:
            print(cfg, prec, "no counter file", r.returncode, r.stderr[-300:])
            continue
        busy, gui, dur = collections.defaultdict(list), collections.defaultdict(list), collections.defaultdict(list)
        name = lambda n: re.sub(r"\(.*", "", n).replace("void ", "")
        for row in csv.DictReader(open(cc[-1])):
            k = name(row["Kernel_Name"])
            if k.startswith(WIDE):
                (busy if row["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES" else gui)[k].append(float(row["Counter_Value"]))
                if "Start_Timestamp" in row and row.get("End_Timestamp"):
                    dur[k].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
        if kt and not any(dur.values()):
            for row in csv.DictReader(open(kt[-1])):
                k = name(row["Kernel_Name"])
                if k.startswith(WIDE):
                    dur[k].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
        for k in sorted(busy):
            b = sum(busy[k]) / len(busy[k])
            us = sum(dur[k]) / max(len(dur[k]), 1) if dur.get(k) else float("nan")
            frac = b / (us * 1e-6 * 2.4e9 * 1024) if us == us and us > 0 else float("nan")
            print(f"{cfg:6s} {prec:7s} {k[:58]:58s} {len(busy[k]):4d} {us:8.1f} {b / 1e6:15.2f} {frac:10.3f}", flush=True)
        for f in glob.glob(os.path.join(d, "**", "*"), recursive=True):      # keep the directory small
            if os.path.isfile(f) and not f.endswith(("counter_collection.csv",)):
                os.remove(f)
