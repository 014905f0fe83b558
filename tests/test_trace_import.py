"""TLC state / trace import (vsr_tlaplus_amd/csrc/vsr_parse.hpp behind vsrmc_model_parse_states).

CPU part: the reader is the inverse of the printer, and the printer is pinned line by line to the reference's
state_transfer_violation_trace.txt (test_host_cpu.py), so reading back the printed golden states must give the golden
records; when /root/reference is present (this container, not the GPU box) the real file is read as well.
GPU part (-m gpu): the 24 states of that trace are a behaviour of the lowered model — every step is a successor generated
by the HIP kernels, under the action TLC names, and the last state violates AcknowledgedWriteNotLost."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

REF_TRACE = "/root/reference/state_transfer_violation_trace.txt"


@pytest.fixture(scope="module")
def vt():
    import __graft_entry__
    __graft_entry__.build()
    import vsr_tlaplus_amd as vt
    return vt


def _model(vt, p):
    return vt.Model.from_constants(R=p["R"], C_=p["C"], n=len(p["values"]), L=p["L"])


def _rec(st):
    return np.array([int(w, 16) for w in st["words"]], dtype=np.uint64)


def _normal(m, rec):
    h0 = int(m.layout.fixed_words)                                      # wire layout: header + replica blocks, then the bag
    r = np.array(rec, dtype=np.uint64)
    r[h0:] = np.sort(r[h0:])
    return r


def _trace_expression(m, states):
    """the golden states printed in the layout of the reference's file (trace:1-30)"""
    blocks = []
    for st in states:
        body = m.format_state(_rec(st)).splitlines()[1:-1]
        head = ' _TEAction |-> [\n   position |-> %d,\n   name |-> "%s",\n   location |-> "Unknown location"\n ],' % (
            int(st["position"]), st["action"])
        blocks.append("[\n" + head + "\n" + "\n".join(body) + "\n]")
    return "<<\n" + ",\n".join(blocks) + "\n>>\n"


def test_reader_inverts_the_printer_on_the_golden_trace(vt, golden_trace):
    m = _model(vt, golden_trace["params"])
    for st in golden_trace["states"]:
        got = m.parse_states(m.format_state(_rec(st)))
        assert len(got) == 1 and got[0][0] is None
        assert np.array_equal(got[0][1], _normal(m, _rec(st))), st["position"]


def test_reader_takes_a_trace_expression_and_the_console_form(vt, golden_trace):
    m = _model(vt, golden_trace["params"])
    states = golden_trace["states"]
    got = m.parse_states(_trace_expression(m, states))
    assert [a for a, _ in got] == [st["action"] for st in states]
    for (_, rec), st in zip(got, states):
        assert np.array_equal(rec, _normal(m, _rec(st)))
    # TLC's console form: "State k: <Action line ...>" + /\ var = value
    text = ""
    for st in states[:6]:
        body = [l.rstrip(",") for l in m.format_state(_rec(st)).splitlines()[1:-1]]
        name = st["action"] if st["action"] != "Initial predicate" else "Initial predicate"
        text += "State %d: <%s line 1, col 1 to line 2, col 2 of module VSR>\n" % (int(st["position"]), name)
        text += "\n".join("/\\ " + l.replace(" |-> ", " = ", 1) for l in body) + "\n\n"
    got = m.parse_states(text)
    assert [a for a, _ in got] == [st["action"] for st in states[:6]]
    for (_, rec), st in zip(got, states):
        assert np.array_equal(rec, _normal(m, _rec(st)))


@pytest.mark.skipif(not os.path.exists(REF_TRACE), reason="the reference checkout is not on this machine")
def test_reader_on_the_reference_file_itself(vt, golden_trace):
    m = _model(vt, golden_trace["params"])
    with open(REF_TRACE) as f:
        got = m.parse_states(f.read())
    assert len(got) == 24
    for (action, rec), st in zip(got, golden_trace["states"]):
        assert action == st["action"]
        assert np.array_equal(rec, _normal(m, _rec(st))), st["position"]


def test_reader_refuses_what_the_packed_record_cannot_hold(vt, golden_trace):
    m = _model(vt, golden_trace["params"])
    good = m.format_state(_rec(golden_trace["states"][7]))
    for bad, what in [(good.replace("rep_view_number |-> <<", "rep_view_number |-> <<9, "), "ReplicaCount"),
                      (good.replace("aux_svc |-> ", "aux_svc |-> 9"), "outside the range"),
                      (good.replace("rep_status |-> <<Normal", "rep_status |-> <<Recovering"), "Recovering"),
                      (good.replace("aux_svc", "aux_bogus"), "unknown variable"),
                      (good.replace("operation |-> v1", "operation |-> v9"), "Values"),
                      (good[: len(good) // 2], "expected"),
                      ("", "empty")]:
        with pytest.raises(vt.VsrmcError) as ei:
            m.parse_states(bad)
        assert what in str(ei.value), (what, str(ei.value))
    # variables the text leaves out keep their Init value
    only = m.parse_states("[ rep_view_number |-> <<1, 1, 1>> ]")
    assert np.array_equal(only[0][1], m.init_state())


def test_reader_round_trips_deep_config2_states(vt):
    """NewState messages with first_op > 1 (function-valued logs), DVC slots, acked values: the 28-state counter-example"""
    with open(os.path.join(GOLDEN, "config2_violation.json")) as f:
        fx = json.load(f)
    m = vt.Model.from_constants(R=3, C_=1, n=2, L=2)
    for st in fx["trace"]:
        rec = np.array([int(w, 16) for w in st["words"]], dtype=np.uint64)
        got = m.parse_states(m.format_state(rec))
        assert np.array_equal(got[0][1], _normal(m, rec))


@pytest.mark.gpu
def test_reference_trace_is_a_behaviour_of_the_hip_successor_function(vt, golden_trace):
    m = _model(vt, golden_trace["params"])
    states = golden_trace["states"]
    parsed = m.parse_states(_trace_expression(m, states))
    res = m.check_trace([rec for _, rec in parsed])
    assert res["ok"] and res["first_bad"] == -1
    assert res["actions"] == [st["action"] for st in states]
    assert res["inv_mask_last"] == 1                                     # AcknowledgedWriteNotLost
    # the ordinals replay to the same states
    tr = m.replay(res["ords"])
    assert [a for a, _ in tr] == res["actions"]
    for (_, rec), (_, want) in zip(tr, parsed):
        assert np.array_equal(_normal(m, rec), want)
    # a corrupted behaviour is caught at the right place
    broken = [rec.copy() for _, rec in parsed]
    broken[10] = broken[9]
    res = m.check_trace(broken)
    assert not res["ok"] and res["first_bad"] == 10
    res = m.check_trace(broken[1:])
    assert not res["ok"] and res["first_bad"] == 0


@pytest.mark.gpu
def test_cli_validate_trace(vt, golden_trace, tmp_path):
    m = _model(vt, golden_trace["params"])
    from test_host_cpu import _cfg
    cfg = _cfg(tmp_path, R=3, vals="v1, v2, v3", L=3)                              # README:13-18
    tf = tmp_path / "trace.txt"
    tf.write_text(_trace_expression(m, golden_trace["states"]))
    r = subprocess.run([os.path.join(ROOT, "vsr_tlaplus_amd", "vsrmc"), "-config", str(cfg), "-noTLA", "-validateTrace", str(tf)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 12, r.stdout + r.stderr
    assert "24 states read" in r.stdout and "The trace is a behaviour of the model." in r.stdout
    assert "Its last state violates invariant AcknowledgedWriteNotLost." in r.stdout
    assert "State 24: <%s>" % golden_trace["states"][-1]["action"] in r.stdout


@pytest.mark.gpu
def test_cli_dump_is_readable_and_complete(vt, tmp_path):
    """vsrmc -dump writes every distinct state in TLC's -dump text form; the product's reader takes it back and the set of
    fingerprints is the oracle's whole state space of config 1 (76 states) — the file a maintainer with a JVM would diff
    against `tlc2.TLC -dump` to pin parity with TLC itself."""
    from oracle import orc
    from test_host_cpu import _cfg
    cfg = _cfg(tmp_path, R=2, vals="v1", L=1)
    out = tmp_path / "states.dump"
    r = subprocess.run([os.path.join(ROOT, "vsr_tlaplus_amd", "vsrmc"), "-config", cfg, "-noTLA", "-tableLog2", "16", "-frontierGiB", "0.01",
                        "-dump", str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "76 states dumped" in r.stdout, r.stdout + r.stderr
    text = out.read_text()
    assert text.startswith("State 1:\n/\\ aux_client_acked = <<>>\n") and text.count("\nState ") == 75
    m = vt.Model.from_constants(R=2, C_=1, n=1, L=1)
    states = m.parse_states(text)
    assert len(states) == 76 and np.array_equal(states[0][1], m.init_state())
    words = np.concatenate([rec for _, rec in states])
    off = np.cumsum([0] + [len(rec) for _, rec in states]).astype(np.uint64)
    fps, _ = m.fingerprints(words, off)
    P = orc.Params(2, 1, 1, 1)
    ob = orc.Bfs(P)
    want = []
    level = 1
    while True:
        want += [int(x) for x in ob.level_fps(level)]
        if ob.step() == 0:
            break
        level += 1
    assert sorted(int(x) for x in fps) == sorted(want) and len(set(want)) == 76


@pytest.mark.gpu
def test_diff_tlc_dump_tool(vt, tmp_path):
    """tools/diff_tlc_dump.py on a stand-in for `tlc2.TLC -dump`: this checker's own dump of (2, {v1,v2}, 1) — 163 states under
    VIEW + SYMMETRY — with the states shuffled and the two values swapped in every second one (TLC keeps whichever representative it
    meets first): the sets are equal.  With one state removed the tool names it; with a state of another configuration added, too."""
    import random
    import re
    import sys
    from test_host_cpu import _cfg
    cfg = _cfg(tmp_path, R=2, vals="v1, v2", L=1)
    out = tmp_path / "states.dump"
    r = subprocess.run([os.path.join(ROOT, "vsr_tlaplus_amd", "vsrmc"), "-config", cfg, "-noTLA", "-tableLog2", "16", "-frontierGiB", "0.05",
                        "-dump", str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "163 states dumped" in r.stdout, r.stdout + r.stderr
    blocks = [b for b in re.split(r"(?m)^State \d+:\n", out.read_text()) if b.strip()]
    assert len(blocks) == 163
    swap = lambda t: re.sub(r"\bv([12])\b", lambda mm: "v2" if mm.group(1) == "1" else "v1", t)   # noqa: E731
    blocks = [swap(b) if k % 2 else b for k, b in enumerate(blocks)]
    random.Random(5).shuffle(blocks)
    tool = [sys.executable, os.path.join(ROOT, "tools", "diff_tlc_dump.py"), "-config", cfg, "--table-log2", "16", "--frontier-gib", "0.05"]

    def run(bl, *extra):
        f = tmp_path / "tlc.dump"
        f.write_text("".join("State %d:\n%s" % (k + 1, b) for k, b in enumerate(bl)))
        return subprocess.run(tool + [str(f)] + list(extra), capture_output=True, text=True, timeout=300)
    r = run(blocks)
    assert r.returncode == 0 and "163 distinct under VIEW + SYMMETRY" in r.stdout and "The two sets of states are equal." in r.stdout, r.stdout + r.stderr
    r = run(blocks[:-1])
    assert r.returncode == 1 and "only in the GPU BFS: 1" in r.stdout and "reached by:" in r.stdout, r.stdout + r.stderr
    assert run(blocks[:-1], "--subset-ok").returncode == 0
    alien = blocks[0].replace("aux_svc = 0", "aux_svc = 0").replace("rep_view_number = <<1, 1>>", "rep_view_number = <<3, 3>>")
    if alien != blocks[0]:
        r = run(blocks + [alien])
        assert r.returncode == 1 and "only in the TLC dump: 1" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_cli_dump_trace_tla_round_trip(vt, tmp_path):
    """vsrmc -dumpTrace tla FILE writes the counter-example in the form of the reference's golden file (a TLA+ trace expression with
    _TEAction records); -validateTrace reads it back, re-walks it on the GPU and confirms the violation; the test-side TLC value parser
    reads the same file."""
    from oracle import tlcvalue
    from test_host_cpu import _cfg
    cfg = _cfg(tmp_path, L=1, extra="AcknowledgedWritesExistOnMajority")          # (3,1,{v1,v2},1): violated at depth 19
    out = tmp_path / "cex.tla.txt"
    cli = os.path.join(ROOT, "vsr_tlaplus_amd", "vsrmc")
    r = subprocess.run([cli, "-config", cfg, "-noTLA", "-tableLog2", "20", "-frontierGiB", "0.1", "-dumpTrace", "tla", str(out)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 12 and "The counter-example was written to" in r.stdout, r.stdout[-2000:] + r.stderr
    text = out.read_text()
    assert text.startswith("<<\n[\n _TEAction |-> [\n   position |-> 1,\n   name |-> \"Initial predicate\",") and text.rstrip().endswith(">>")
    states = tlcvalue.parse_value(text)
    assert len(states) == 19 and [dict(dict(s)["_TEAction"])["position"] for s in states] == list(range(1, 20))
    r2 = subprocess.run([cli, "-config", cfg, "-noTLA", "-validateTrace", str(out)], capture_output=True, text=True, timeout=300)
    assert "19 states read" in r2.stdout and "The trace is a behaviour of the model." in r2.stdout, r2.stdout + r2.stderr
    assert "Its last state violates invariant AcknowledgedWritesExistOnMajority." in r2.stdout
