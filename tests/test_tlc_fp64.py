"""The TLC-style fingerprint mode (vsr_tlaplus_amd/csrc/vsr_tlcfp.hpp behind vsrmc_tlc_* ).

Nothing here is checked against TLC itself — no JVM, no TLC jar, no pinned fingerprint exists (every TLC fact is [TLC-RECALLED]).  What is pinned:
 * the Rabin arithmetic: the product's byte table = the oracle's bit-serial register = a polynomial division over GF(2) written here with Python ints;
 * the serialiser: the product walks the packed record, the oracle (oracle/tlc_fp64.cpp) builds a generic value tree from the unpacked state and sorts
   it with a generic comparison — the byte streams must be identical;
 * the order of record fields, set elements and function domains: the byte stream, parsed back generically and printed in TLC's syntax, reads exactly as
   the states of the reference's own TLC output (state_transfer_violation_trace.txt, through the printer that test_host_cpu.py pins to it line by line);
 * SYMMETRY: which permuted state TLC fingerprints (the smallest by compareTo over all variables in declaration order) — the oracle compares whole value
   trees generically, the product compares the packed record under two permutations; the same permutation must come out for every state;
 * (-m gpu) the kernel = the oracle, fingerprint by fingerprint, and FP64 separates the states of whole levels."""
import json
import os
import struct

import numpy as np
import pytest

from conftest import GOLDEN

IRRED = 0x911498AE0E66BAD6
MASK = (1 << 64) - 1


@pytest.fixture(scope="module")
def vt():
    import __graft_entry__
    __graft_entry__.build()
    import vsr_tlaplus_amd as vt
    return vt


@pytest.fixture(scope="module")
def orc():
    from oracle import orc
    return orc


# ---- a third statement of FP64: polynomials over GF(2) as Python ints (bit i = coefficient of x^i), plain long division --------------------------
def _reflect(v):
    return int("{:064b}".format(v)[::-1], 2)


P_POLY = (1 << 64) | _reflect(IRRED)          # the irreducible polynomial of degree 64: the register holds bit 63 = x^0


def _polymod(a):
    while a.bit_length() > 64:
        a ^= P_POLY << (a.bit_length() - 65)
    return a


def py_fp64(data, start=IRRED):
    """FP64 of `data` continued from register `start`: the register is a polynomial r (reflected); a byte b (bit k = coefficient of x^(63-k) after the
    xor into the low byte) gives r' = (r + b) * x^8 mod P."""
    r = _reflect(start)
    for b in data:
        bp = 0
        for k in range(8):
            if (b >> k) & 1:
                bp |= 1 << (63 - k)
        r = _polymod((r ^ bp) << 8)
    return _reflect(r)


def _ext(vt, fp, data):
    from vsr_tlaplus_amd import capi
    buf = np.frombuffer(bytes(data), dtype=np.uint8) if len(data) else np.zeros(1, dtype=np.uint8)
    return int(capi.load().vsrmc_fp64_extend(fp, buf.ctypes.data, len(data)))


def test_fp64_three_statements_agree(vt, orc):
    from vsr_tlaplus_amd import capi
    assert int(capi.load().vsrmc_fp64_new()) == IRRED
    rng = np.random.default_rng(64)
    cases = [b"", b"\x00", b"\x01", b"\x80", b"a", b"view_number", bytes(range(256))] + [rng.integers(0, 256, int(n), dtype=np.uint8).tobytes() for n in (1, 7, 8, 9, 63, 64, 65, 1000, 4182)]
    for data in cases:
        want = py_fp64(data)
        assert _ext(vt, IRRED, data) == want, data[:16]
        assert orc.fp64_bytes(data) == want, data[:16]
    assert py_fp64(b"") == IRRED


def test_fp64_is_a_rabin_fingerprint(vt):
    """properties no table typo survives: affine over GF(2) for equal lengths, continuation = concatenation, one flipped bit always changes it"""
    rng = np.random.default_rng(65)
    for n in (1, 5, 64, 777):
        a = rng.integers(0, 256, n, dtype=np.uint8)
        b = rng.integers(0, 256, n, dtype=np.uint8)
        z = np.zeros(n, dtype=np.uint8)
        f = lambda d: _ext(vt, IRRED, d.tobytes())
        assert f(a ^ b) ^ f(z) == f(a) ^ f(b)
        k = int(rng.integers(0, n + 1))
        assert _ext(vt, _ext(vt, IRRED, a[:k].tobytes()), a[k:].tobytes()) == f(a)
        for _ in range(20):
            c = a.copy()
            c[int(rng.integers(0, n))] ^= 1 << int(rng.integers(0, 8))
            assert f(c) != f(a)
    # the register is 64 bits of remainder: a string of the reflected polynomial's own bytes after 8 zero-bytes-worth of shifting is not special-cased
    assert _ext(vt, 0, b"\x00" * 100) == 0                                    # the zero polynomial stays zero (New() != 0 is what makes lengths count)
    assert _ext(vt, IRRED, b"\x00") != _ext(vt, IRRED, b"\x00\x00")


# ---- generic reader of the byte stream: tags only, no knowledge of the model ------------------------------------------------------------------------
class _Rd:
    def __init__(self, data):
        self.d, self.p = data, 0

    def byte(self):
        self.p += 1
        return self.d[self.p - 1]

    def int(self):
        self.p += 4
        return struct.unpack_from("<i", self.d, self.p - 4)[0]

    def value(self):
        t = self.byte()
        if t == 0:
            return ("bool", {ord("t"): True, ord("f"): False}[self.byte()])
        if t == 1:
            return ("int", self.int())
        if t == 21:
            return ("model", self.int())
        if t == 3:
            n = self.int()
            self.p += n
            return ("str", self.d[self.p - n:self.p].decode())
        if t == 5:
            return ("set", [self.value() for _ in range(self.int())])
        if t == 9:
            n = self.int()
            return ("fcn", [(self.value(), self.value()) for _ in range(n)])
        raise AssertionError("unknown tag %d at %d" % (t, self.p - 1))


def _model_values(n):
    return ["v%d" % (k + 1) for k in range(n)] + ["Normal", "ViewChange", "Recovering", "RequestMsg", "ReplyMsg", "PrepareMsg", "PrepareOkMsg", "CommitMsg",
                                                  "StartViewChangeMsg", "DoViewChangeMsg", "StartViewMsg", "GetStateMsg", "NewStateMsg", "RecoveryMsg",
                                                  "RecoveryResponseMsg", "Nil"]          # creation order = VSR.cfg:6-24


def _show(v, names):
    """TLC's toString of a value, as far as this model's values go"""
    k, x = v
    if k == "bool":
        return "TRUE" if x else "FALSE"
    if k == "int":
        return str(x)
    if k == "model":
        return names[x]
    if k == "set":
        if x and all(e[0] == "int" for e in x):
            vals = [e[1] for e in x]
            assert vals == list(range(vals[0], vals[0] + len(vals)))
            return "%d..%d" % (vals[0], vals[-1])
        return "{" + ", ".join(_show(e, names) for e in x) + "}"
    assert k == "fcn"
    if x and all(d[0] == "str" for d, _ in x):
        return "[" + ", ".join("%s |-> %s" % (d[1], _show(r, names)) for d, r in x) + "]"
    if all(d[0] == "int" for d, _ in x) and [d[1] for d, _ in x] == list(range(1, len(x) + 1)):
        return "<<" + ", ".join(_show(r, names) for _, r in x) + ">>"
    return "(" + " @@ ".join("%s :> %s" % (_show(d, names), _show(r, names)) for d, r in x) + ")"


MSG_TYPES = ("StartViewChangeMsg", "PrepareMsg", "PrepareOkMsg", "DoViewChangeMsg", "StartViewMsg", "GetStateMsg", "NewStateMsg")
VIEW_NAMES = [["rep_status", "rep_log", "rep_view_number", "rep_op_number", "rep_peer_op_number", "rep_commit_number", "rep_client_table", "rep_last_normal_view"],
              ["rep_rec_number", "rep_rec_recv"], ["rep_svc_recv", "rep_dvc_recv", "rep_sent_dvc", "rep_sent_sv"], []]    # VSR.tla:140-150


def _view_as_text(stream, n_values):
    rd = _Rd(stream)
    k, view = rd.value()
    assert rd.p == len(stream) and k == "fcn" and [d for d, _ in view] == [("int", i) for i in range(1, 8)]
    names = _model_values(n_values)
    out = {}
    for group, (_, val) in zip(VIEW_NAMES, view[:4]):
        assert val[0] == "fcn" and [d for d, _ in val[1]] == [("int", i) for i in range(1, len(group) + 1)]
        for name, (_, v) in zip(group, val[1]):
            out[name] = _show(v, names)
    out["replicas"], out["clients"], out["messages"] = (_show(view[k][1], names) for k in (4, 5, 6))
    return out


def _printed(m, rec):
    lines = {}
    for ln in m.format_state(rec).splitlines()[1:-1]:
        name, val = ln.rstrip(",").split(" |-> ", 1)
        lines[name] = val
    return lines


def _rec(st):
    return np.array([int(w, 16) for w in st["words"]], dtype=np.uint64)


def test_stream_reads_as_the_reference_trace(vt, orc, golden_trace):
    """the 24 states TLC printed for the reference (README config, three values): fields, set elements and message-function domains come in the stream in
    the order TLC printed them; the oracle's stream is the same bytes; all six permutations give parseable, distinct-or-equal streams of equal length"""
    p = golden_trace["params"]
    n = len(p["values"])
    m = vt.Model.from_constants(R=p["R"], C_=p["C"], n=n, L=p["L"])
    P = orc.Params(p["R"], p["C"], n, p["L"])
    types = set()
    for st in golden_trace["states"]:
        rec = _rec(st)
        stream = m.tlc_view_bytes(rec, 0)
        assert stream == orc.tlc_view_bytes(P, rec, 0)
        got, want = _view_as_text(stream, n), _printed(m, rec)
        assert set(got) == {k for k in want if not k.startswith("aux_")}
        for name in got:
            assert got[name] == want[name], (st["position"], name)
        types.update(t for t in MSG_TYPES if t in got["messages"])
        for perm in range(int(m.layout.permutations)):
            s2 = m.tlc_view_bytes(rec, perm)
            assert len(s2) == len(stream) and s2 == orc.tlc_view_bytes(P, rec, perm), (st["position"], perm)
        fp, perm = orc.tlc_fingerprint(P, rec, with_perm=True)
        assert perm == m.tlc_min_permutation(rec) and fp == py_fp64(m.tlc_view_bytes(rec, perm)), st["position"]
    assert len(types) == 6, types                    # every live message type but NewStateMsg (next test)


@pytest.mark.parametrize("fixture", ["config2_violation.json", "config3_violation.json"])
def test_streams_along_the_counterexamples(vt, orc, fixture):
    """the 28- and 24-state counter-examples of the BFS (state transfer: GetState and NewState messages with their interval-domain log functions)"""
    g = json.load(open(os.path.join(GOLDEN, fixture)))
    p = g["params"]
    n = p["n"] if "n" in p else len(p["values"])
    m = vt.Model.from_constants(R=p["R"], C_=p["C"], n=n, L=p["L"])
    P = orc.Params(p["R"], p["C"], n, p["L"])
    types = set()
    for k, st in enumerate(g["trace"]):
        rec = _rec(st)
        for perm in range(int(m.layout.permutations)):
            assert m.tlc_view_bytes(rec, perm) == orc.tlc_view_bytes(P, rec, perm), (k, perm)
        assert orc.tlc_fingerprint(P, rec, with_perm=True)[1] == m.tlc_min_permutation(rec), k
        got, want = _view_as_text(m.tlc_view_bytes(rec, 0), n), _printed(m, rec)
        for name in got:
            assert got[name] == want[name], (k, name)
        types.update(t for t in MSG_TYPES if t in got["messages"])
    assert len(types) == (7 if fixture.startswith("config2") else 6), types          # NewStateMsg: in the 28-state trace


@pytest.mark.parametrize("cfg,depth", [((3, 1, 1, 1), 40), ((3, 1, 2, 2), 8), ((3, 1, 3, 1), 9), ((3, 2, 2, 1), 7), ((5, 1, 2, 1), 6)])
def test_streams_of_product_and_oracle_are_identical(vt, orc, cfg, depth):
    """every state of the first levels (the whole state space of the one-value configuration: state transfer, view changes, every message type), every
    permutation: packed-record walker = value-tree serialiser, byte for byte; the stream parses back and prints as the product's TLC printer prints it"""
    R, C_, n, L = cfg
    P = orc.Params(R, C_, n, L, assume_commit_number=(C_ > 1))          # VSR.tla:421 is an evaluation error with two clients
    m = vt.Model.from_constants(R=R, C_=C_, n=n, L=L, assume_commit_number=(C_ > 1))
    b = orc.Bfs(P)
    total, types, chosen = 0, set(), {}
    for _ in range(depth):
        w, off = b.frontier()
        for i in range(len(off) - 1):
            rec = w[int(off[i]):int(off[i + 1])]
            for perm in range(int(m.layout.permutations)):
                assert m.tlc_view_bytes(rec, perm) == orc.tlc_view_bytes(P, rec, perm), (total, perm)
            fp, perm = orc.tlc_fingerprint(P, rec, with_perm=True)           # the permuted state TLC fingerprints: generic compareTo over the value trees
            assert perm == m.tlc_min_permutation(rec), total                 # = the packed-record comparison of the product
            chosen[perm] = chosen.get(perm, 0) + 1
            if total % 5 == 0:                                               # the other members of the state's symmetry class: the same fingerprint from each
                for q in range(1, int(m.layout.permutations)):
                    rq = orc.tlc_permute_record(P, rec, q)
                    fq, pq = orc.tlc_fingerprint(P, rq, with_perm=True)
                    assert fq == fp, (total, q)
                    assert pq == m.tlc_min_permutation(rq), (total, q)
                    chosen[pq] = chosen.get(pq, 0) + 1
            if total % 37 == 0:
                got, want = _view_as_text(m.tlc_view_bytes(rec, 0), n), _printed(m, rec)
                for name in got:
                    assert got[name] == want[name], (total, name)
                types.update(t for t in MSG_TYPES if t in got["messages"])
            total += 1
        if b.step() <= 0:
            break
    assert total > 500
    assert len(chosen) == int(m.layout.permutations), chosen          # every permutation is TLC's choice for some state
    if cfg == (3, 1, 1, 1):
        assert len(types) == 5, types            # the whole state space of this configuration: everything but state transfer (the golden trace covers that)


def test_without_symmetry_one_permutation(vt, orc):
    P = orc.Params(3, 1, 2, 2, symmetry=False)
    m = vt.Model.from_constants(R=3, C_=1, n=2, L=2, symmetry=False)
    assert int(m.layout.permutations) == 1
    rec = orc.init_record(P)
    assert orc.tlc_fingerprint(P, rec) == py_fp64(m.tlc_view_bytes(rec, 0))


def test_other_models_are_refused(vt):
    m2 = vt.Model.second_model()
    with pytest.raises(Exception):
        m2.tlc_view_bytes(np.zeros(16, dtype=np.uint64), 0)


# ---- GPU ---------------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("cfg,depth,sym", [((3, 1, 2, 2), 11, True), ((3, 1, 3, 1), 9, True), ((3, 1, 2, 2), 9, False), ((5, 1, 2, 1), 6, True)])
def test_kernel_fingerprints_equal_the_oracle(vt, orc, cfg, depth, sym):
    """k_tlc_fingerprints over the oracle's own frontier records = the oracle's fingerprint (value tree, bit-serial FP64, min over permutations), state by state"""
    R, C_, n, L = cfg
    P = orc.Params(R, C_, n, L, symmetry=sym)
    m = vt.Model.from_constants(R=R, C_=C_, n=n, L=L, symmetry=sym)
    b = orc.Bfs(P)
    total = 0
    for _ in range(depth):
        w, off = b.frontier()
        got = m.tlc_fingerprints(w, off)
        step = max(1, (len(off) - 1) // 3000)                      # the oracle's tree + bit-serial division is slow: every state of small levels, a stride of big ones
        variants, want = [], []
        for i in range(0, len(off) - 1, step):
            rec = w[int(off[i]):int(off[i + 1])]
            fp = orc.tlc_fingerprint(P, rec)
            assert int(got[i]) == fp, (total, i)
            if sym and total % 7 == 0:                             # another member of the symmetry class must come out of the kernel with the same fingerprint
                variants.append(orc.tlc_permute_record(P, rec, 1 + total % (int(m.layout.permutations) - 1)))
                want.append(fp)
            total += 1
        if variants:
            voff = np.concatenate([[0], np.cumsum([len(v) for v in variants])]).astype(np.uint64)
            assert [int(x) for x in m.tlc_fingerprints(np.concatenate(variants), voff)] == want
        if b.step() <= 0:
            break
    assert total > 2000


@pytest.mark.gpu
def test_fp64_separates_the_states_of_whole_levels(vt):
    """the shipped configuration through level 17 (3.2 M states of that level): FP64 from the frontier in HBM gives as many distinct values as the level has
    states, level by level and across levels — the mode is a second, independent identity of the same state classes"""
    m = vt.Model.from_constants(R=3, C_=1, n=2, L=2)
    mc = vt.ModelChecker(m, table_log2=24, frontier_words=1 << 27, frontier_states=1 << 22, pending_entries=1 << 24, keep_trace=False)
    seen = []
    total = 0
    while mc.level < 17:
        info = mc.step()
        fps, ms = mc.tlc_level_fps()
        assert len(fps) == info["n_new"] and len(np.unique(fps)) == info["n_new"], mc.level
        assert not np.any(fps == 0)
        seen.append(fps)
        total += info["n_new"]
    assert total == 3340498 - 1                                             # levels 2-17 of the shipped configuration (oracle_levels_config2.json)
    allfp = np.concatenate(seen)
    assert len(np.unique(allfp)) == total
    mc.close()
