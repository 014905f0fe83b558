#!/usr/bin/env python3
"""rocprofv3 --pmc result databases of the two TCC passes of tools/profile_round.sh -> profiles/<tag>_<workload>_traffic.json:
HBM-side bytes of all k_expand launches of one run, per launch, from the L2's fabric request counters by request size
(TCC_EA0_RDREQ_{32B,64B,128B}, TCC_EA0_WRREQ / _64B) — no calibration factor — with the sha256 of the kernel sources they were
measured on (bench.py shows the figure only for a build of exactly these sources).

    python tools/make_traffic.py TAG WORKLOAD RD.db WR.db "COMMAND" > profiles/TAG_WORKLOAD_traffic.json"""
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def sums(path):
    db = sqlite3.connect(path)
    out, calls = {}, 0
    for name, counter, v, n in db.execute("select name, counter_name, sum(counter_value), count(*) from pmc_events group by name, counter_name"):
        if "k_expand" in name:
            out[counter] = out.get(counter, 0.0) + v
    for name, n in db.execute("select name, total_calls from top_kernels"):
        if "k_expand" in name:
            calls += n
    return out, calls


def main():
    tag, workload, rd_db, wr_db, cmd = sys.argv[1:6]
    import bench
    rd, calls = sums(rd_db)
    wr, calls_w = sums(wr_db)
    n32, n64, n128, nrd = (rd.get(k, 0.0) for k in ("TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum", "TCC_EA0_RDREQ_sum"))
    w64, nwr = wr.get("TCC_EA0_WRREQ_64B_sum", 0.0), wr.get("TCC_EA0_WRREQ_sum", 0.0)
    other = nrd - n32 - n64 - n128                               # requests of a size no counter names: tallied at 64 B, reported
    rd_bytes = 32 * n32 + 64 * n64 + 128 * n128 + 64 * max(other, 0.0)
    wr_bytes = 64 * w64 + 32 * (nwr - w64)
    print(json.dumps(dict(
        source="rocprofv3 --kernel-trace --pmc (two passes: TCC_EA0_RDREQ by size; TCC_EA0_WRREQ, _64B) -- %s (%s)" % (cmd, tag),
        workload=workload, kernel="k_expand (every instantiation launched by the run)", launches=int(calls),
        rdreq=dict(n32=n32, n64=n64, n128=n128, total=nrd, unsized=other), wrreq=dict(n64=w64, total=nwr),
        atomics_to_fabric=wr.get("TCC_EA0_ATOMIC_sum"), read_bytes=rd_bytes, write_bytes=wr_bytes,
        hbm_bytes_per_launch=round((rd_bytes + wr_bytes) / max(1, calls)), kernel_source_sha256=bench.kernel_source_sha256()), indent=1))


if __name__ == "__main__":
    main()
