"""oracle/pycodec.py — packed-record codec for pyoracle states (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

Third, independent statement of the packed format v1 (DESIGN.md "Packed record"): python value <-> list of
u64 words.  Used by tests to move states between the golden trace / pyoracle and the C++ oracle / HIP path.
"""
from . import pyoracle as po

TYPE_CODE = {po.StartViewChangeMsg: 1, po.PrepareMsg: 2, po.PrepareOkMsg: 3, po.DoViewChangeMsg: 4,
             po.StartViewMsg: 5, po.GetStateMsg: 6, po.NewStateMsg: 7}
CODE_TYPE = {v: k for k, v in TYPE_CODE.items()}
STATUS_CODE = {po.Normal: 0, po.ViewChange: 1, po.Recovering: 2}
CODE_STATUS = {v: k for k, v in STATUS_CODE.items()}


def wpr(M):
    return 1 + (M.R + 2) // 2


def fixed_words(M):
    return 1 + M.R * wpr(M)


def enc_entry(M, e):
    d = dict(e)
    return d["view_number"] | (M.Values.index(d["operation"]) << 3) | ((d["client_id"] - 1) << 5) | (d["request_number"] << 6)


def dec_entry(M, b):
    return po.rec(view_number=b & 7, operation=M.Values[(b >> 3) & 3], client_id=((b >> 5) & 1) + 1,
                  request_number=(b >> 6) & 3)


def enc_seq_log(M, log):
    w = 0
    for i, e in enumerate(log):
        w |= enc_entry(M, e) << (8 * i)
    return w


def dec_seq_log(M, w):
    out = []
    for i in range(3):
        b = (w >> (8 * i)) & 0xFF
        if b:
            out.append(dec_entry(M, b))
    return tuple(out)


def enc_msg(M, m, count):
    d = dict(m)
    t = TYPE_CODE[d["type"]]
    lg = 0
    if t == 2:
        lg = enc_entry(M, d["message"])
    elif t in (4, 5):
        lg = enc_seq_log(M, d["log"])
    elif t == 7:
        for on, e in d["log"]:
            lg |= enc_entry(M, e) << (8 * (on - 1))
    return (t | (d["view_number"] << 3) | (d["dest"] << 6) | (d["source"] << 9) | (d.get("op_number", 0) << 12) |
            (d.get("commit_number", 0) << 14) | (d.get("last_normal_vn", 0) << 16) | (d.get("first_op", 0) << 19) |
            (count << 21) | (lg << 32))


def dec_msg(M, w):
    t = w & 7
    view, dest, src = (w >> 3) & 7, (w >> 6) & 7, (w >> 9) & 7
    op, commit, lnv, first = (w >> 12) & 3, (w >> 14) & 3, (w >> 16) & 7, (w >> 19) & 3
    count = (w >> 21) & 3
    lg = w >> 32
    typ = CODE_TYPE[t]
    if t == 1:
        m = po.rec(type=typ, view_number=view, dest=dest, source=src)
    elif t == 2:
        m = po.rec(type=typ, view_number=view, message=dec_entry(M, lg & 0xFF), op_number=op, commit_number=commit,
                   dest=dest, source=src)
    elif t in (3, 6):
        m = po.rec(type=typ, view_number=view, op_number=op, dest=dest, source=src)
    elif t == 4:
        m = po.rec(type=typ, view_number=view, log=dec_seq_log(M, lg), last_normal_vn=lnv, op_number=op,
                   commit_number=commit, dest=dest, source=src)
    elif t == 5:
        m = po.rec(type=typ, view_number=view, log=dec_seq_log(M, lg), op_number=op, commit_number=commit,
                   dest=dest, source=src)
    else:
        log = tuple((on, dec_entry(M, (lg >> (8 * (on - 1))) & 0xFF)) for on in range(first, op + 1))
        m = po.rec(type=typ, view_number=view, log=log, first_op=first, op_number=op, commit_number=commit,
                   dest=dest, source=src)
    return m, count


def pack(M, s):
    """pyoracle state -> list of u64 words (bag in canonical python order)."""
    msgs = sorted(s["messages"].items(), key=lambda kv: po.canon(kv[0]))
    hdr = len(msgs) | (s["aux_svc"] << 8)
    for i, v in enumerate(M.Values):
        if v in s["aux_client_acked"]:
            hdr |= (2 if s["aux_client_acked"][v] else 1) << (11 + 2 * i)
    words = [hdr]
    for r in range(1, M.R + 1):
        i = r - 1
        A = (STATUS_CODE[s["rep_status"][i]] | (s["rep_view_number"][i] << 2) | (s["rep_op_number"][i] << 5) |
             (s["rep_commit_number"][i] << 7) | (s["rep_last_normal_view"][i] << 9) |
             (int(s["rep_sent_dvc"][i]) << 12) | (int(s["rep_sent_sv"][i]) << 13))
        for m in s["rep_svc_recv"][i]:
            d = dict(m)
            assert d["dest"] == r and d["view_number"] == s["rep_view_number"][i], "I1"
            A |= 1 << (14 + d["source"] - 1)
        for p in range(M.R):
            A |= s["rep_peer_op_number"][i][p] << (19 + 2 * p)
        for c in range(M.C):
            row = dict(s["rep_client_table"][i][c])
            A |= (row["request_number"] | (row["op_number"] << 2) | (int(row["executed"]) << 4)) << (29 + 5 * c)
        assert s["rep_op_number"][i] == len(s["rep_log"][i]), "I3"
        x = [0] * (2 * (wpr(M) - 1))
        x[0] = enc_seq_log(M, s["rep_log"][i])
        for m in s["rep_dvc_recv"][i]:
            d = dict(m)
            assert d["dest"] == r and d["view_number"] == s["rep_view_number"][i] and x[d["source"]] == 0, "I2"
            x[d["source"]] = 1 | (d["last_normal_vn"] << 1) | (d["op_number"] << 4) | (d["commit_number"] << 6) | (
                enc_seq_log(M, d["log"]) << 8)
        words.append(A)
        for k in range(wpr(M) - 1):
            words.append(x[2 * k] | (x[2 * k + 1] << 32))
    for m, c in msgs:
        words.append(enc_msg(M, m, c))
    return words


def unpack(M, words):
    """list of u64 words -> pyoracle state."""
    hdr = words[0]
    nmsg = hdr & 0xFF
    s = po.Init(M)
    s["aux_svc"] = (hdr >> 8) & 7
    acked = {}
    for i, v in enumerate(M.Values):
        a = (hdr >> (11 + 2 * i)) & 3
        if a:
            acked[v] = (a == 2)
    s["aux_client_acked"] = acked
    W = wpr(M)
    cols = {k: [] for k in ("rep_status", "rep_view_number", "rep_op_number", "rep_commit_number",
                            "rep_last_normal_view", "rep_sent_dvc", "rep_sent_sv", "rep_svc_recv",
                            "rep_peer_op_number", "rep_client_table", "rep_log", "rep_dvc_recv")}
    for r in range(1, M.R + 1):
        b = words[1 + (r - 1) * W: 1 + r * W]
        A = b[0]
        view = (A >> 2) & 7
        cols["rep_status"].append(CODE_STATUS[A & 3])
        cols["rep_view_number"].append(view)
        cols["rep_op_number"].append((A >> 5) & 3)
        cols["rep_commit_number"].append((A >> 7) & 3)
        cols["rep_last_normal_view"].append((A >> 9) & 7)
        cols["rep_sent_dvc"].append(bool((A >> 12) & 1))
        cols["rep_sent_sv"].append(bool((A >> 13) & 1))
        cols["rep_svc_recv"].append(frozenset(
            po.rec(type=po.StartViewChangeMsg, view_number=view, dest=r, source=src)
            for src in range(1, M.R + 1) if (A >> (14 + src - 1)) & 1))
        cols["rep_peer_op_number"].append(tuple((A >> (19 + 2 * p)) & 3 for p in range(M.R)))
        rows = []
        for c in range(M.C):
            row = (A >> (29 + 5 * c)) & 31
            rows.append(po.rec(request_number=row & 3, op_number=(row >> 2) & 3, executed=bool((row >> 4) & 1)))
        cols["rep_client_table"].append(tuple(rows))
        x = []
        for k in range(W - 1):
            x.append(b[1 + k] & 0xFFFFFFFF)
            x.append(b[1 + k] >> 32)
        cols["rep_log"].append(dec_seq_log(M, x[0]))
        dv = set()
        for src in range(1, M.R + 1):
            if x[src] & 1:
                dv.add(po.rec(type=po.DoViewChangeMsg, view_number=view, log=dec_seq_log(M, x[src] >> 8),
                              last_normal_vn=(x[src] >> 1) & 7, op_number=(x[src] >> 4) & 3,
                              commit_number=(x[src] >> 6) & 3, dest=r, source=src))
        cols["rep_dvc_recv"].append(frozenset(dv))
    for k, v in cols.items():
        s[k] = tuple(v)
    msgs = {}
    for w in words[fixed_words(M): fixed_words(M) + nmsg]:
        m, c = dec_msg(M, w)
        msgs[m] = c
    s["messages"] = msgs
    return s


def normalise(M, words):
    """sort the bag words so that records can be compared as lists"""
    f = fixed_words(M)
    n = words[0] & 0xFF
    return list(words[:f]) + sorted(words[f:f + n])
