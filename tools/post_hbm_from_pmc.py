#!/usr/bin/env python3
"""Achieved HBM GB/s per post-process kernel from the two PMC passes (tools/pmc_analyze.py outputs of FETCH_SIZE and WRITE_SIZE,
KB per launch; gfx950 FETCH_SIZE half-count correction: bytes = 2*FETCH + WRITE): post_hbm_from_pmc.py fetch.txt write.txt"""
import re, sys
def parse(path, key):
    out, cur = {}, None
    for line in open(path):
        m = re.match(r"(\S.*?)\s+grid (\d+)\s+n=(\d+) dur (\d+) us", line)
        if m:
            cur = (m.group(1).strip(), m.group(2)); out.setdefault(cur, {})["dur"] = float(m.group(4)); continue
        m = re.search(key + r"=([0-9.e+]+)", line)
        if m and cur: out[cur][key] = float(m.group(1))
    return out
f, w = parse(sys.argv[1], "FETCH_SIZE"), parse(sys.argv[2], "WRITE_SIZE")
print("%-28s %-10s %9s %11s %11s %9s" % ("kernel", "grid", "dur us", "read MB", "write MB", "GB/s"))
for k in f:
    if not any(s in k[0] for s in ("binarize", "ccl_", "gather", "trace_", "contour", "compact", "suffix", "pool_off", "pack_u8", "dilate")):
        continue
    rd = 2 * f[k].get("FETCH_SIZE", 0.0) / 1024; wr = w.get(k, {}).get("WRITE_SIZE", 0.0) / 1024
    dur = f[k]["dur"]
    print("%-28s %-10s %9.0f %11.2f %11.2f %9.0f" % (k[0].replace("ptocr::", "")[:28], k[1], dur, rd, wr, (rd + wr) / max(dur, 1e-9) * 1e3))     # MB / us = TB/s
