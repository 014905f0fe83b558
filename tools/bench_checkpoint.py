#!/usr/bin/env python3
"""Checkpoint / recover at scale: config 2 up to level 25 (123 758 810 distinct states), save, destroy, recover into a table of
a different size, run on to the depth-28 violation; prints the file size and the save / load times."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vsr_tlaplus_amd as vt  # noqa: E402

path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/vsrmc_c2_l25.chk"
m = vt.Model.from_constants(R=3, C_=1, n=2, L=2)
sizes = dict(frontier_words=1 << 32, frontier_states=1 << 27, pending_entries=1 << 20, trace_entries=1 << 29)
mc = vt.ModelChecker(m, table_log2=30, **sizes)
while mc.level < 25:
    mc.step()
t0 = time.perf_counter()
mc.save(path)
t_save = time.perf_counter() - t0
distinct = mc.distinct
mc.close()
size = os.path.getsize(path)
t0 = time.perf_counter()
mc = vt.ModelChecker(m, table_log2=31, recover=path, **sizes)
t_load = time.perf_counter() - t0
assert (mc.level, mc.distinct) == (25, distinct)
t0 = time.perf_counter()
why = mc.run()
t_run = time.perf_counter() - t0
assert why == "violation" and mc.distinct == 319228361 and mc.violation["fp"] == 0x22239cb457b78204
tr = mc.trace(mc.violation["level"], mc.violation["index"])
assert len(tr) == 28
os.remove(path)
print(json.dumps(dict(level=25, distinct_at_save=distinct, file_GB=round(size / 1e9, 2), save_s=round(t_save, 2), load_s=round(t_load, 2),
                      save_GBps=round(size / 1e9 / t_save, 2), load_GBps=round(size / 1e9 / t_load, 2), rest_of_run_s=round(t_run, 3))))
