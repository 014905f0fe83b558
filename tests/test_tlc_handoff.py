"""The TLC hand-off kit (tools/tlc_handoff.sh + tools/diff_tlc_dump.py) is known to work before anyone with a JVM tries it: the differ is
fed a dump in TLC's `-dump` syntax that no part of the product wrote — the Python restatement's states printed by oracle/tlcprint.py —
and must find the GPU BFS equal to it; with one state removed / one foreign state added it must say which."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dump(tmp_path, R, values, L, drop=None):
    from oracle import pyoracle, tlcprint
    M = pyoracle.Model(R=R, C=1, values=values, L=L)
    levels, _, _ = pyoracle.bfs(M)
    flat = [s for lv in levels for s in lv]
    if drop is not None:
        flat = flat[:drop] + flat[drop + 1:]
    path = tmp_path / ("py_%d_%d.dump" % (R, len(values)))
    path.write_text(tlcprint.dump([flat]))
    return str(path), len(flat)


def test_tlc_syntax_printer_round_trips_through_the_tlc_value_parser():
    from oracle import pyoracle, tlcprint, tlcvalue
    M = pyoracle.Model(R=2, C=1, values=("v1", "v2"), L=2)
    levels, _, _ = pyoracle.bfs(M, max_depth=12)
    for s in [x for lv in levels for x in lv][::7]:
        for var in pyoracle.VIEW_VARS + pyoracle.AUX_VARS:
            back = tlcvalue.parse_value(tlcprint.fmt(s[var]))
            if back == () and s[var] == {}:                      # TLC prints the empty function as <<>>
                continue
            assert pyoracle.canon(back) == pyoracle.canon(s[var]), (var, tlcprint.fmt(s[var]))


@pytest.mark.gpu
def test_differ_accepts_a_dump_the_product_did_not_write(tmp_path):
    from test_host_cpu import _cfg
    cfg = _cfg(tmp_path, R=2, vals="v1, v2", L=2)
    dump, n = _dump(tmp_path, 2, ("v1", "v2"), 2)
    assert n == 2073
    tool = [sys.executable, os.path.join(ROOT, "tools", "diff_tlc_dump.py"), "-config", cfg]
    r = subprocess.run(tool + [dump], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "The two sets of states are equal." in r.stdout and "2073 states, 2073 distinct" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    # one state missing from the dump: the GPU BFS has one more, --subset-ok accepts, the strict comparison does not
    dump2, _ = _dump(tmp_path, 2, ("v1", "v2"), 2, drop=1000)
    r = subprocess.run(tool + [dump2], capture_output=True, text=True, timeout=600)
    assert r.returncode == 1 and "only in the GPU BFS: 1" in r.stdout, r.stdout[-2000:]
    r = subprocess.run(tool + [dump2, "--subset-ok"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "Every TLC state was found by the GPU BFS." in r.stdout
    # a state of ANOTHER configuration in the dump (three replicas' worth of limit): reported as only in the dump
    cfg1 = _cfg(tmp_path, R=2, vals="v1, v2", L=1)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "diff_tlc_dump.py"), "-config", cfg1, dump], capture_output=True, text=True, timeout=600)
    assert r.returncode == 1 and "The sets DIFFER." in r.stdout and "only in the TLC dump: " in r.stdout
