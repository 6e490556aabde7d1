"""Turn a rocpd PMC summary (tools/rocpd_summary.py output of the FETCH_SIZE and WRITE_SIZE passes of bench.py) into
profiles/pmc_traffic.json, which bench.py reads to fill roofline.traffic.
Correction (MI355X_MICROARCH.md, HBM section): on gfx950 FETCH_SIZE counts 64 B per 128-B request -> x2; the factor
is re-derived here from the 64 MiB device-to-device copy that the same trace contains (its byte count is known).
usage: python tools/pmc_to_json.py <pmc_summary.txt> <round-tag>"""
import json
import re
import sys

txt = open(sys.argv[1]).read().splitlines()
rows = []
for l in txt:
    m = re.match(r"^(.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+([\d.]+)\s+(\d+)\s+(\d+)\s+(\d+)\s*$", l)
    if m:
        rows.append((m.group(1).strip(), m.group(2), float(m.group(3)), int(m.group(4))))
copy_fetch = max(v for n, c, v, g in rows if "copyBuffer" in n and c == "FETCH_SIZE")
copy_write = max(v for n, c, v, g in rows if "copyBuffer" in n and c == "WRITE_SIZE")
KNOWN_COPY_KB = 65536.0           # the 64 MiB frame clone bench.py makes before the timed region
fcorr, wcorr = KNOWN_COPY_KB / copy_fetch, KNOWN_COPY_KB / copy_write
eng_fetch = [v for n, c, v, g in rows if n.startswith("pl_engine") and c == "FETCH_SIZE"][0]
eng_write = [v for n, c, v, g in rows if n.startswith("pl_engine") and c == "WRITE_SIZE"][0]
out = {
    "source": sys.argv[1], "round": sys.argv[2], "kernel": "pl_engine", "workload": "4096x4096 RGBA8 s=19 b=2, one launch",
    "FETCH_SIZE_KB_raw": eng_fetch, "WRITE_SIZE_KB_raw": eng_write,
    "fetch_correction": round(fcorr, 4), "write_correction": round(wcorr, 4),
    "calibration": "64 MiB copyBuffer in the same trace: FETCH_SIZE %.1f KB, WRITE_SIZE %.1f KB" % (copy_fetch, copy_write),
    "traffic_bytes": int((eng_fetch * fcorr + eng_write * wcorr) * 1024),
    "algorithmic_bytes": 8 * 4096 * 4096,
}
json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
