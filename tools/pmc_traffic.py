"""Turn the text summaries of the FETCH_SIZE / WRITE_SIZE rocprofv3 passes (tools/rocpd_summary.py output) into the
per-launch HBM traffic record bench.py reports as roofline.traffic.

    python tools/pmc_traffic.py WORKLOAD KERNEL_SUBSTRING fetch_summary.txt write_summary.txt out.json

FETCH_SIZE / WRITE_SIZE count kilobytes; on gfx950 FETCH_SIZE under-reports by 2x (MI355X_MICROARCH.md, HBM / rocprofv3
section), so bytes = (2 * fetch_kb + write_kb) * 1024."""
import json
import os
import re
import sys


def pmc_avg(path, counter, kernel):
    for line in open(path):
        m = re.match(r"\s*PMC (.*?)\s+" + counter + r"\s+n=\s*(\d+) avg=([0-9.eE+]+)", line)
        if m and kernel in m.group(1):
            return float(m.group(3)), int(m.group(2))
    raise SystemExit(f"{path}: no {counter} row for a kernel matching {kernel!r}")


def main():
    workload, kernel, fetch_txt, write_txt, out = sys.argv[1:6]
    fetch_kb, nf = pmc_avg(fetch_txt, "FETCH_SIZE", kernel)
    write_kb, nw = pmc_avg(write_txt, "WRITE_SIZE", kernel)
    rec = {}
    if os.path.exists(out):
        rec = json.load(open(out))
    rec[workload] = {"kernel": kernel, "fetch_kb": fetch_kb, "write_kb": write_kb, "launches": [nf, nw],
                     "bytes": int((2 * fetch_kb + write_kb) * 1024)}
    json.dump(rec, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(rec[workload]))


if __name__ == "__main__":
    main()
