#!/usr/bin/env python3
"""Achieved HBM GB/s per kernel from the two PMC passes (tools/pmc_analyze.py outputs of FETCH_SIZE and WRITE_SIZE, KB per launch;
gfx950 FETCH_SIZE half-count correction: bytes = 2*FETCH + WRITE), and their sum per call as JSON for bench.py's `traffic`:
    post_hbm_from_pmc.py fetch.txt write.txt out.json [post|bf16]"""
import json, re, sys
def parse(path, key):
    out, cur = {}, None
    for line in open(path):
        m = re.match(r"(\S.*?)\s+grid (\d+)\s+n=(\d+) dur (\d+) us", line)
        if m:
            cur = (m.group(1).strip(), m.group(2)); out.setdefault(cur, {})["dur"] = float(m.group(4)); out[cur]["n"] = int(m.group(3)); continue
        m = re.search(key + r"=([0-9.e+]+)", line)
        if m and cur: out[cur][key] = float(m.group(1))
    return out
f, w = parse(sys.argv[1], "FETCH_SIZE"), parse(sys.argv[2], "WRITE_SIZE")
mode = sys.argv[4] if len(sys.argv) > 4 else "post"
POST = ("binarize", "ccl_", "select_starts", "border_states", "scatter_states", "border_stage", "border_wave", "border_quad", "contour", "compact", "suffix", "pool_off", "pack_u8", "dilate")
BF16 = ("bf16", "se_fc")
print("%-34s %-10s %6s %9s %11s %11s %9s" % ("kernel", "grid", "n", "dur us", "read MB", "write MB", "GB/s"))
total_b, calls, per = 0.0, None, {}
for k in f:
    if not any(s in k[0] for s in (POST if mode == "post" else BF16)):
        continue
    rd = 2 * f[k].get("FETCH_SIZE", 0.0) / 1024; wr = w.get(k, {}).get("WRITE_SIZE", 0.0) / 1024
    dur, n = f[k]["dur"], f[k]["n"]
    name = k[0].replace("ptocr::", "").replace("void ", "")[-34:]
    print("%-34s %-10s %6d %9.0f %11.2f %11.2f %9.0f" % (name, k[1], n, dur, rd, wr, (rd + wr) / max(dur, 1e-9) * 1e3))     # MB / us = TB/s
    total_b += (rd + wr) * n * 1e6
    e = per.setdefault(name, {"launches": 0, "mb": 0.0, "us": 0.0}); e["launches"] += n; e["mb"] += (rd + wr) * n; e["us"] += dur * n
# calls = launches of a once-per-call kernel
once = [v["launches"] for kk, v in per.items() if ("pool_offsets" in kk or "stem3x3" in kk)]
calls = once[0] if once else 1
for v in per.values():
    v["gbps"] = round(v["mb"] / max(v["us"], 1e-9) * 1e3, 1); v["mb_per_call"] = round(v["mb"] / calls, 2); v["us_per_call"] = round(v["us"] / calls, 1)
    del v["mb"], v["us"]
print("calls %d, HBM bytes per call %.1f MB" % (calls, total_b / calls / 1e6))
key = "hbm_bytes_per_call" if mode == "post" else "hbm_bytes_per_forward"
json.dump({key: total_b / calls, "per_kernel": per, "note": "2*FETCH_SIZE + WRITE_SIZE per launch (gfx950 FETCH_SIZE half-count correction), rocprofv3 --pmc, separate passes"},
          open(sys.argv[3], "w"), indent=1)
