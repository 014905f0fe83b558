"""CPU-side tests (`-m "not gpu"`): the C-ABI library loads and exports every symbol include/vsrmc.h declares, the cfg
reader / model lowering / TLC-format printer (host logic, no compute calls), and the oracle against the counter-example
fixture the GPU run produced."""
import ctypes as C
import hashlib
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
VSR_CFG_TEXT = """\\* (test template in the layout of a TLC cfg: comment lines, blank lines, multi-line INVARIANT list)

CONSTANTS
    ReplicaCount = %(R)d
    ClientCount = %(C)d
    Values = {%(vals)s}
    StartViewOnTimerLimit = %(L)d
    RestartEmptyLimit = %(restart)d
    Normal = Normal
    ViewChange = ViewChange
    Recovering = Recovering
    RequestMsg = RequestMsg
    ReplyMsg = ReplyMsg
    PrepareMsg = PrepareMsg
    PrepareOkMsg = PrepareOkMsg
    CommitMsg = CommitMsg
    StartViewChangeMsg = StartViewChangeMsg
    DoViewChangeMsg = DoViewChangeMsg
    StartViewMsg = StartViewMsg
    GetStateMsg = GetStateMsg
    NewStateMsg = NewStateMsg
    RecoveryMsg = RecoveryMsg
    RecoveryResponseMsg = RecoveryResponseMsg
    Nil = Nil

INIT Init
NEXT Next

VIEW view
\\* symmetry reduction needs at least two model values
%(symmetry)s

\\* (no temporal properties: safety checking only)

INVARIANT
AcknowledgedWriteNotLost
\\* AcknowledgedWritesExistOnMajority \\* kept out by default, enabled by some tests through the line below
%(extra)s
"""


def _cfg(tmp_path, R=3, C_=1, vals="v1, v2", L=2, restart=0, symmetry="SYMMETRY symmValues", extra="\\* NoLogDivergence"):
    p = tmp_path / "VSR.cfg"
    p.write_text(VSR_CFG_TEXT % dict(R=R, C=C_, vals=vals, L=L, restart=restart, symmetry=symmetry, extra=extra))
    return str(p)


@pytest.fixture(scope="module")
def vt():
    import __graft_entry__
    __graft_entry__.build()
    import vsr_tlaplus_amd as vt
    return vt


def test_library_exports_every_declared_symbol(vt):
    hdr = open(os.path.join(ROOT, "include", "vsrmc.h")).read()
    declared = set(re.findall(r"\b(vsrmc_[a-z_0-9]+)\s*\(", hdr))
    from vsr_tlaplus_amd import capi
    assert declared == set(capi.SYMBOLS), declared ^ set(capi.SYMBOLS)
    lib = C.CDLL(capi.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert vt.load().vsrmc_version() >= 100


def test_no_gpu_means_loud_failure_not_fallback(vt):
    """The product path has no CPU fallback: without a HIP device every compute entry point fails with VSRMC_E_HIP."""
    if vt.load().vsrmc_device_count() > 0:
        pytest.skip("a GPU is visible here")
    m = vt.Model.from_constants()
    with pytest.raises(vt.VsrmcError) as ei:
        vt.FPSet()
    assert ei.value.code == -3
    with pytest.raises(vt.VsrmcError) as ei:
        vt.ModelChecker(m)
    assert ei.value.code == -3
    w = m.init_state()
    with pytest.raises(vt.VsrmcError) as ei:
        m.get_next_states(w, np.array([0, len(w)], dtype=np.uint64))
    assert ei.value.code == -3


def test_product_sources_never_touch_the_oracle():
    pkg = os.path.join(ROOT, "vsr_tlaplus_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", ".java", ".c")):
                text = open(os.path.join(dp, f)).read()
                assert "oracle" not in text.replace("CPU oracle", "").replace("the oracle", "") or f == "__none__", (dp, f)


def test_cfg_reader_accepts_the_shipped_grammar(vt, tmp_path):
    m = vt.Model.load(_cfg(tmp_path))
    lay = m.layout
    assert (lay.replica_count, lay.client_count, lay.value_count, lay.start_view_on_timer_limit) == (3, 1, 2, 2)
    assert lay.symmetry == 1 and lay.invariant_mask == 1 and lay.permutations == 2
    assert lay.words_per_replica == 3 and lay.fixed_words == 10
    m3 = vt.Model.load(_cfg(tmp_path, vals="v1, v2, v3", L=3))          # the README defect config
    assert m3.layout.value_count == 3 and m3.layout.permutations == 6
    m1 = vt.Model.load(_cfg(tmp_path, R=2, vals="v1", L=1, symmetry="\\* SYMMETRY symmValues"))
    assert m1.layout.symmetry == 0 and m1.layout.permutations == 1
    m5 = vt.Model.load(_cfg(tmp_path, R=5, extra="AcknowledgedWritesExistOnMajority"))
    assert m5.layout.invariant_mask == 3 and m5.layout.words_per_replica == 4 and m5.layout.fixed_words == 21


@pytest.mark.parametrize("kw,needle", [
    (dict(restart=1), "RestartEmptyLimit"),
    (dict(R=7), "supported bounds"),
    (dict(extra="NoSuchInvariant"), "unknown INVARIANT"),
    (dict(symmetry="SYMMETRY other"), "SYMMETRY"),
    (dict(symmetry="PROPERTY Liveness"), "PROPERTY"),
    (dict(symmetry="SPECIFICATION Spec"), "SPECIFICATION"),
    (dict(vals=""), "supported bounds"),
])
def test_cfg_reader_rejects_what_it_cannot_honour(vt, tmp_path, kw, needle):
    with pytest.raises(vt.VsrmcError) as ei:
        vt.Model.load(_cfg(tmp_path, **kw))
    assert ei.value.code == -2 and needle in ei.value.message


def test_unknown_module_is_refused(vt, tmp_path):
    other = tmp_path / "Other.tla"
    other.write_text("---- MODULE Other ----\n====\n")
    with pytest.raises(vt.VsrmcError) as ei:
        vt.Model.load(_cfg(tmp_path), str(other))
    assert ei.value.code == -2 and "sha256" in ei.value.message


@pytest.mark.skipif(not os.path.exists("/root/reference/vsr-revisited/paper/VSR.tla"), reason="reference not mounted")
def test_reference_cfg_and_module_load_as_is(vt):
    m = vt.Model.load("/root/reference/vsr-revisited/paper/VSR.cfg", "/root/reference/vsr-revisited/paper/VSR.tla")
    assert (m.layout.replica_count, m.layout.value_count, m.layout.start_view_on_timer_limit) == (3, 2, 2)


def test_init_state_matches_oracle(vt):
    from oracle import orc
    for (R, C_, n, L) in [(2, 1, 1, 1), (3, 1, 2, 2), (3, 2, 3, 3), (5, 1, 2, 2)]:
        m = vt.Model.from_constants(R=R, C_=C_, n=n, L=L)
        assert np.array_equal(m.init_state(), orc.init_record(orc.Params(R, C_, n, L)))


def test_tlc_printer_reproduces_the_reference_trace_text(vt, golden_trace):
    """Every `var |-> value` line our printer emits for the 24 golden states has the SHA-256 of the corresponding line of
    /root/reference/state_transfer_violation_trace.txt (digests in the fixture; the text itself is not copied)."""
    p = golden_trace["params"]
    m = vt.Model.from_constants(R=p["R"], C_=p["C"], n=len(p["values"]), L=p["L"])
    for st in golden_trace["states"]:
        rec = np.array([int(w, 16) for w in st["words"]], dtype=np.uint64)
        text = m.format_state(rec)
        lines = [l.rstrip(",") for l in text.splitlines()[1:-1]]
        got = {l.split(" |-> ", 1)[0]: hashlib.sha256(l.encode()).hexdigest() for l in lines}
        for var, dig in st["line_sha256"].items():
            assert got[var] == dig, (st["position"], var)
        # the three variables the (older) reference trace does not have are constant in every BASELINE config
        assert "aux_restart |-> 0" in lines and any(l.startswith("rep_rec_number |-> <<0") for l in lines)


def test_tlc_printer_output_parses_back_to_the_same_state(vt):
    from oracle import pycodec, pyoracle as po, tlcvalue
    with open(os.path.join(GOLDEN, "config2_violation.json")) as f:
        fx = json.load(f)
    M = po.Model(3, 1, ("v1", "v2"), 2)
    m = vt.Model.from_constants(R=3, C_=1, n=2, L=2)
    for st in fx["trace"][::3] + fx["trace"][-1:]:
        words = [int(w, 16) for w in st["words"]]
        val = tlcvalue.parse_value(m.format_state(np.array(words, dtype=np.uint64)))
        state = {k: v for k, v in dict(val).items()}
        want = pycodec.unpack(M, words)
        empty = lambda x: {} if x == () else x     # `<<>>` is the empty function as well as the empty tuple  # noqa: E731

        def seq_logs(msgs):
            """NewStateMsg.log is a function first_op..op_number -> entry (VSR.tla:535-536); with first_op = 1 that IS a sequence and
            TLC prints it as one (<<e1, ..>>), which parses back as a tuple: compare both forms as (op number, entry) pairs"""
            out = {}
            for m, c in (msgs.items() if isinstance(msgs, dict) else []):
                d = dict(m)
                if d.get("type") == "NewStateMsg" and d["log"] and not isinstance(d["log"][0][0], int):
                    d["log"] = tuple((d["first_op"] + i, e) for i, e in enumerate(d["log"]))
                out[tuple(sorted(d.items()))] = c
            return out
        for k in want:
            a, b = (seq_logs(state[k]), seq_logs(want[k])) if k == "messages" else (state[k], want[k])
            assert po.canon(empty(a)) == po.canon(empty(b)), k


def test_config2_counterexample_is_a_behaviour(golden_counts):
    """The 28-state counter-example the GPU BFS found for the shipped VSR.cfg constants (3 replicas, {v1,v2}, limit 2):
    both CPU restatements accept every step, the named action produces it, and AcknowledgedWriteNotLost
    (VSR.tla:945-950) holds in states 1..27 and fails in state 28."""
    from oracle import orc, pycodec, pyoracle as po
    with open(os.path.join(GOLDEN, "config2_violation.json")) as f:
        fx = json.load(f)
    P = orc.Params(3, 1, 2, 2)
    M = po.Model(3, 1, ("v1", "v2"), 2)
    recs = [np.array([int(w, 16) for w in t["words"]], dtype=np.uint64) for t in fx["trace"]]
    norm = lambda w: tuple(int(x) for x in orc.normalise(P, w))   # noqa: E731
    assert len(recs) == fx["depth"] == 28
    assert norm(recs[0]) == norm(orc.init_record(P))
    for i in range(len(recs) - 1):
        hits = [s for s in orc.successors(P, recs[i]) if norm(s["words"]) == norm(recs[i + 1])]
        assert hits and orc.ACTIONS[hits[0]["action"]] == fx["trace"][i + 1]["action"], i
        cur = pycodec.unpack(M, [int(x) for x in recs[i]])
        nxt = pycodec.normalise(M, [int(x) for x in recs[i + 1]])
        names = [n for n, t in po.successors(M, cur) if pycodec.normalise(M, pycodec.pack(M, t)) == nxt]
        assert fx["trace"][i + 1]["action"] in names, i
    assert [orc.invariants(P, r) for r in recs] == [0] * 27 + [1]
    assert not po.AcknowledgedWriteNotLost(M, pycodec.unpack(M, [int(x) for x in recs[-1]]))
    fp, _ = orc.fingerprint(P, recs[-1])
    assert "%016x" % fp == fx["viol_fp"]
    # the fixture's level counts agree with the oracle's own fixture as deep as the oracle went
    g = golden_counts["config2 (3,1,{v1,v2},2)"]
    for lv, mine in zip(g["levels"], fx["levels"]):
        assert (lv["new"], lv["generated"], lv["deadlocks"]) == (mine["n_new"], mine["generated"], mine["deadlocks"])


def test_cli_help_and_cfg_errors(vt, tmp_path):
    cli = os.path.join(ROOT, "vsr_tlaplus_amd", "vsrmc")
    if not os.path.exists(cli):
        pytest.skip("CLI not built")
    r = subprocess.run([cli, "--help"], capture_output=True, text=True)
    assert r.returncode == 0 and "-config" in r.stdout
    r = subprocess.run([cli, "-config", _cfg(tmp_path, restart=2), "VSR.tla", "-noTLA"], capture_output=True, text=True)
    assert r.returncode != 0 and "RestartEmptyLimit" in (r.stderr + r.stdout)


def test_make_cfg_tool_writes_a_cfg_the_loader_accepts(vt, tmp_path):
    import subprocess
    out = tmp_path / "gen.cfg"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_cfg.py"), str(out), "3", "1", "v1, v2, v3", "3"], check=True)
    lay = vt.Model.load(str(out)).layout
    assert (lay.replica_count, lay.client_count, lay.value_count, lay.start_view_on_timer_limit) == (3, 1, 3, 3)
    assert lay.symmetry == 1 and lay.permutations == 6 and lay.invariant_mask == 1


def test_test_hooks_live_in_the_hooks_library_only(vt):
    """Round-4 review (hygiene): the test hook VSRMC_TEST_FORCE_BAD is compiled only with -DVSRMC_TEST_HOOKS — into libvsrmc_hooks.so, which
    vsr_tlaplus_amd/build.py makes beside the product library and only tests/test_sharded_gloo.py loads.  The product library does not even contain
    the variable's name; both export the same C ABI."""
    from vsr_tlaplus_amd import capi
    here = os.path.dirname(capi.LIB_PATH)
    prod = open(os.path.join(here, "libvsrmc.so"), "rb").read()
    hooks = open(os.path.join(here, "libvsrmc_hooks.so"), "rb").read()
    assert b"VSRMC_TEST_FORCE_BAD" not in prod
    assert b"VSRMC_TEST_FORCE_BAD" in hooks
    lib = C.CDLL(os.path.join(here, "libvsrmc_hooks.so"))
    for name in capi.SYMBOLS:
        assert hasattr(lib, name), name
