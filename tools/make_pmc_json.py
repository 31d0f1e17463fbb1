"""rocprofv3 output of tools/profile_bench.sh -> the JSON bench.py's `roofline` reads (stdout) and a readable table (stderr).
Per kernel, per launch: average duration (kernel_stats.csv of the --stats pass), FETCH_SIZE / WRITE_SIZE in bytes (the counters
report KB: x 1024, MI355X_MICROARCH.md), SQ counters summed over the launch; derived: mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES /
(1024 SIMDs x duration x SCLK) with the SCLK taken as SQ_BUSY_CYCLES' own clock = 2.1 GHz nominal under load (stated, not
measured), wait_any = SQ_WAIT_ANY / SQ_WAVE_CYCLES, valu_active = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES,
lds_bank_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE."""
import collections, csv, glob, json, sys

d = sys.argv[1]
# kernels whose reads are wide coalesced streams (16 B per lane): FETCH_SIZE tallies their 128-B requests at 64 B, so the guide
# (MI355X_MICROARCH.md, HBM) says to double it.  k_bin_accumulate: the record stream (three aligned 16-byte loads per lane and run);
# k_render_bwd_t16 / _h3: the x-stash (two float4 per lane).  The forward's reads are scattered 8-byte gathers: uncalibrated, left as read.
FETCH_X2 = ("k_bin_accumulate", "k_render_bwd_t16", "k_render_bwd_h3")
KEEP = ("k_render_fwd", "k_render_bwd", "k_bin_accumulate", "k_bin_scatter", "k_bin_count", "k_src_points", "k_sample_points_grid", "k_reduce_dw", "k_bin_rowscan", "k_loss")


def short(name):   # "void (anonymous namespace)::k_render_bwd_t16<0, false, false, true>(scanerf::BwdArgs)" -> "k_render_bwd_t16<0, false, false, true>"
    n = name.replace("void ", "").replace("(anonymous namespace)::", "")
    depth = 0
    for i, ch in enumerate(n):
        depth += ch == "<"
        depth -= ch == ">"
        if ch == "(" and depth == 0:
            return n[:i].strip()
    return n.strip()


kern = collections.defaultdict(dict)
f = glob.glob(f"{d}/stats/**/*kernel_stats.csv", recursive=True)
if f:
    for r in csv.DictReader(open(f[0])):
        k = short(r["Name"])
        if k.startswith(KEEP):
            kern[k]["avg_us"] = float(r["AverageNs"]) / 1e3
            kern[k]["calls"] = int(r["Calls"])
for f in glob.glob(f"{d}/p*/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k.startswith(KEEP):
            a = acc[(k, r["Counter_Name"])]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    for (k, c), (v, n) in acc.items():
        kern[k][c] = v / n
SCLK = 2.1e9
out = {"_source": "rocprofv3 --kernel-trace --stats and --pmc passes (separate; tools/profile_bench.sh) of `python3 bench.py --gpus 1 --no-cpu-baseline --no-side-legs`: per-launch averages, MI355X, configs[1]",
       "_corrections": "FETCH_SIZE / WRITE_SIZE: reported KB x 1024 (MI355X_MICROARCH.md); FETCH_SIZE tallies the 128-B requests of 16-B/lane streams as 64 B: traffic_bytes = fetch_bytes x 2 + write_bytes for the kernels marked fetch_x2 (record stream of the accumulate, x-stash of the backward), fetch_bytes + write_bytes otherwise; SQ_* are sums over the launch; mfma_busy assumes SCLK 2.1 GHz under load",
       "kernels": {}}
for k, v in sorted(kern.items(), key=lambda kv: -kv[1].get("avg_us", 0)):
    e = {"avg_us": v.get("avg_us"), "calls": v.get("calls")}
    if "FETCH_SIZE" in v:
        e["fetch_bytes"] = v["FETCH_SIZE"] * 1024
    if "WRITE_SIZE" in v:
        e["write_bytes"] = v["WRITE_SIZE"] * 1024
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        e["fetch_x2"] = k.startswith(FETCH_X2)
        e["traffic_bytes"] = e["fetch_bytes"] * (2 if e["fetch_x2"] else 1) + e["write_bytes"]
    for c in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_SALU",
              "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT",
              "SQ_WAIT_INST_LDS", "SQ_WAIT_INST_ANY", "SQ_BUSY_CYCLES"):
        if c in v:
            e[c] = v[c]
    for c, key in (("TCC_EA0_RDREQ_sum", "fabric_read_requests"), ("TCC_EA0_RDREQ_32B_sum", "fabric_read_requests_32B"),
                   ("TCC_EA0_WRREQ_sum", "fabric_write_requests"), ("TCC_EA0_WRREQ_64B_sum", "fabric_write_requests_64B"),
                   ("TCC_REQ_sum", "l2_requests"), ("TCC_HIT_sum", "l2_hits"), ("TCC_MISS_sum", "l2_misses"), ("TCP_TCC_WRITE_REQ_sum", "l1_to_l2_write_requests")):
        if c in v:
            e[key] = v[c]
    if "fabric_read_requests" in e and "fabric_write_requests" in e and v.get("avg_us"):
        # requests between the L2s and the fabric (Infinity Cache / HBM side), reads (64 B, or 128 B tallied as one) + writes (32 / 64 B);
        # the chip serves ~57 G of them per second whatever their size (gather / record-stream micro-benchmarks)
        e["fabric_requests"] = e["fabric_read_requests"] + e["fabric_write_requests"]
        e["fabric_request_rate_Gps"] = e["fabric_requests"] / (v["avg_us"] * 1e-6) / 1e9
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and v.get("avg_us"):
        e["mfma_busy"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * v["avg_us"] * 1e-6 * SCLK)
    if "SQ_WAIT_ANY" in v and v.get("SQ_WAVE_CYCLES"):
        e["wait_any"] = v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"]
    if "SQ_ACTIVE_INST_VALU" in v and v.get("SQ_WAVE_CYCLES"):
        e["valu_active"] = v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"]
    if "SQ_LDS_BANK_CONFLICT" in v and v.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_bank_conflict"] = v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"]
    out["kernels"][k] = e
    print(f"{k[:64]:66s} {e.get('avg_us') or 0:9.1f} us  fetch {e.get('fetch_bytes', 0) / 1e9:6.2f} GB  write {e.get('write_bytes', 0) / 1e9:6.2f} GB  "
          f"mfma {e.get('mfma_busy', 0):.3f}  wait {e.get('wait_any', 0):.3f}  valu {e.get('valu_active', 0):.3f}  ldsconf {e.get('lds_bank_conflict', 0):.3f}  "
          f"fabric req {e.get('fabric_requests', 0):.3g} ({e.get('fabric_request_rate_Gps', 0):.1f} G/s)  "
          f"VALU/MFMA/LDS/VMEM insts {v.get('SQ_INSTS_VALU', 0):.3g}/{v.get('SQ_INSTS_MFMA', 0):.3g}/{v.get('SQ_INSTS_LDS', 0):.3g}/{v.get('SQ_INSTS_VMEM', 0):.3g}", file=sys.stderr)
json.dump(out, sys.stdout, indent=1)
