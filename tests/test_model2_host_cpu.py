"""Second model, host side (no GPU): the cfg loader accepts VR_STATE_TRANSFER.cfg's grammar (SPECIFICATION Spec, the model's own
constants and invariant names), refuses what is not lowered, Init equals the oracle's, and the printer's output parses back to
the Python restatement's value."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CFG = """\\* SPECIFICATION
CONSTANTS
    ReplicaCount = %(R)d
    Values = {%(vals)s}
    StartViewOnTimerLimit = %(L)d
    NoProgressChangeLimit = %(npl)d
    Normal = Normal
    ViewChange = ViewChange
    StateTransfer = StateTransfer
    PrepareMsg = PrepareMsg
    PrepareOkMsg = PrepareOkMsg
    StartViewChangeMsg = StartViewChangeMsg
    DoViewChangeMsg = DoViewChangeMsg
    StartViewMsg = StartViewMsg
    GetStateMsg = GetStateMsg
    NewStateMsg = NewStateMsg
    Nil = Nil
    AnyDest = AnyDest

SPECIFICATION %(spec)s \\* Use when not doing liveness checking

VIEW view
%(extra)s
INVARIANT
AcknowledgedWritesExistOnMajority
NoLogDivergence
CommitNumberNeverHigherThanOpNumber
"""


@pytest.fixture(scope="module")
def vt():
    import __graft_entry__
    __graft_entry__.build()
    import vsr_tlaplus_amd as vt
    return vt


def _cfg(tmp_path, R=3, vals="v1, v2", L=2, npl=0, spec="Spec", extra=""):
    p = tmp_path / ("m2_%d_%d_%d_%s.cfg" % (R, L, npl, spec))
    p.write_text(CFG % dict(R=R, vals=vals, L=L, npl=npl, spec=spec, extra=extra))
    return str(p)


def test_loader_reads_the_second_models_cfg(vt, tmp_path):
    lay = vt.Model.load(_cfg(tmp_path)).layout                     # no .tla: the cfg's constants identify the module
    assert (lay.replica_count, lay.client_count, lay.value_count, lay.start_view_on_timer_limit) == (3, 0, 2, 2)
    assert (lay.symmetry, lay.permutations, lay.invariant_mask, lay.words_per_replica, lay.fixed_words) == (0, 1, 14, 1, 4)
    ref = "/root/reference/vsr-revisited/paper/analysis/03-state-transfer"
    if os.path.exists(ref):                                         # the shipped files themselves, module SHA-256-pinned
        lay = vt.Model.load(ref + "/VR_STATE_TRANSFER.cfg", ref + "/VR_STATE_TRANSFER.tla").layout
        assert (lay.replica_count, lay.value_count, lay.start_view_on_timer_limit, lay.invariant_mask) == (3, 2, 2, 14)
        with pytest.raises(vt.VsrmcError):                          # a cfg of one module with the other module's .tla
            vt.Model.load(ref + "/VR_STATE_TRANSFER.cfg", "/root/reference/vsr-revisited/paper/VSR.tla")


def test_loader_refuses_what_is_not_lowered(vt, tmp_path):
    for kw in (dict(npl=1), dict(spec="LivenessSpec"), dict(extra="SYMMETRY symmValues"), dict(extra="PROPERTY ConvergenceToView"), dict(R=7)):
        with pytest.raises(vt.VsrmcError):
            vt.Model.load(_cfg(tmp_path, **kw))
    with pytest.raises(vt.VsrmcError):
        vt.Model.second_model(no_progress_limit=2)


def test_init_and_printer_against_the_restatements(vt):
    from oracle import orc2, pyoracle2 as po, tlcvalue
    for R, n, L in ((3, 2, 2), (2, 1, 1), (5, 3, 3)):
        m = vt.Model.second_model(R=R, n=n, L=L)
        P = orc2.Params(R, n, L)
        assert np.array_equal(m.init_state(), orc2.init_record(P))
    # states of a small BFS of the Python restatement, printed by the product and parsed back with the test-side TLC value parser
    M = po.Model(3, ("v1", "v2"), 2)
    m = vt.Model.second_model(R=3, n=2, L=2)
    levels, _, _ = po.bfs(M, max_depth=6)
    empty = lambda x: {} if x == () else x                          # noqa: E731

    def seq_logs(msgs):
        out = {}
        for mm, c in (msgs.items() if isinstance(msgs, dict) else []):
            d = dict(mm)
            if d.get("type") == "NewStateMsg" and d["log"] and not isinstance(d["log"][0][0], int):
                d["log"] = tuple((d["first_op"] + i, e) for i, e in enumerate(d["log"]))
            out[tuple(sorted(d.items()))] = c
        return out
    n = 0
    for lvl in levels:
        for s in lvl[::7]:
            val = dict(tlcvalue.parse_value(m.format_state(np.array(po.pack(M, s), dtype=np.uint64))))
            for k in po.VIEW_VARS + ["aux_svc", "aux_client_acked"]:
                a, b = (seq_logs(val[k]), seq_logs(s[k])) if k == "messages" else (val[k], s[k])
                if k == "replicas":
                    b = frozenset(b)
                assert po.canon(empty(a)) == po.canon(empty(b)), k
            n += 1
    assert n > 50


def test_reader_takes_the_printer_back(vt):
    """format_state -> parse_states is the identity on records (the product's TLC reader knows this model's variables and message
    records: what -validateTrace and tools/diff_tlc_dump.py need), in all three text forms: one state record, a trace expression with
    _TEAction fields, console "State k:" blocks; malformed text is refused with the variable's name."""
    from oracle import pyoracle2 as po
    M = po.Model(3, ("v1", "v2"), 2)
    m = vt.Model.second_model(R=3, n=2, L=2)
    levels, _, _ = po.bfs(M, max_depth=8)
    recs = [np.array(po.pack(M, s), dtype=np.uint64) for lvl in levels for s in lvl[::5]]
    assert len(recs) > 150
    norm = lambda w: po.normalise(M, [int(x) for x in w])          # noqa: E731
    texts = [m.format_state(r) for r in recs]
    for r, t in zip(recs, texts):
        back = m.parse_states(t)
        assert len(back) == 1 and norm(back[0][1]) == norm(r)
    sample = list(range(0, len(recs), 9))
    expr = "<<\n" + ",\n".join("[\n _TEAction |-> [\n   position |-> %d,\n   name |-> \"SendSV\",\n   location |-> \"Unknown location\"\n ],\n%s"
                                  % (k + 1, texts[i].split("\n", 1)[1]) for k, i in enumerate(sample)) + "\n>>\n"
    back = m.parse_states(expr)
    assert [a for a, _ in back] == ["SendSV"] * len(sample) and [norm(w) for _, w in back] == [norm(recs[i]) for i in sample]
    console = ""
    for k, i in enumerate(sample):
        lines = [l.rstrip(",") for l in texts[i].splitlines()[1:-1]]
        console += "State %d: <SendDVC line 1, col 1 to line 2, col 2 of module X>\n" % (k + 1) + "\n".join("/\\ " + l.replace(" |-> ", " = ", 1) for l in lines) + "\n\n"
    back = m.parse_states(console)
    assert [a for a, _ in back] == ["SendDVC"] * len(sample) and [norm(w) for _, w in back] == [norm(recs[i]) for i in sample]
    for bad in (texts[3].replace("rep_status |-> <<", "rep_status |-> <<Recovering, ", 1), texts[3].replace("aux_svc |-> ", "aux_svc |-> 9", 1),
                texts[3].replace("rep_view_number", "rep_view_numbr")):
        with pytest.raises(vt.VsrmcError):
            m.parse_states(bad)
