"""oracle/pyoracle.py — SECOND, INDEPENDENT CPU restatement of VSR.tla (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

Written directly from /root/reference/vsr-revisited/paper/VSR.tla (cited as VSR.tla:NNN), not from the C++
oracle: states are plain Python values shaped like the TLA+ values (records = tuples of (field, value)
pairs, sets = frozensets, functions = dicts / tuples), symmetry reduction picks the minimum *value* over
all permutations of Values (no hashing), and VIEW identity is value equality of the `view` tuple
(VSR.tla:149-150).  Its job: pin the C++ oracle (oracle/vsr_oracle.cpp) on whole small state spaces and on
the reference's golden trace (state_transfer_violation_trace.txt) — two independent readings that agree.

"parity unpinned" against TLC itself (no JVM / TLC in this image).
Slow on purpose: pure-Python loops, small cases only.
"""
from itertools import permutations

Normal, ViewChange, Recovering = "Normal", "ViewChange", "Recovering"          # VSR.tla:99-101
PrepareMsg, PrepareOkMsg = "PrepareMsg", "PrepareOkMsg"                        # VSR.tla:104-115
StartViewChangeMsg, DoViewChangeMsg, StartViewMsg = "StartViewChangeMsg", "DoViewChangeMsg", "StartViewMsg"
GetStateMsg, NewStateMsg = "GetStateMsg", "NewStateMsg"
Nil = "Nil"

ACTIONS = ["TimerSendSVC", "ReceiveHigherSVC", "ReceiveMatchingSVC", "SendDVC", "ReceiveHigherDVC",
           "ReceiveMatchingDVC", "SendSV", "ReceiveSV", "ReceiveClientRequest", "ReceivePrepareMsg",
           "ReceivePrepareOkMsg", "ExecuteOp", "SendGetState", "ReceiveGetState", "ReceiveNewState"]

VIEW_VARS = ["rep_status", "rep_log", "rep_view_number", "rep_op_number", "rep_peer_op_number",
             "rep_commit_number", "rep_client_table", "rep_last_normal_view",        # rep_state_vars :140-141
             "rep_rec_number", "rep_rec_recv",                                       # rep_rec_vars :142
             "rep_svc_recv", "rep_dvc_recv", "rep_sent_dvc", "rep_sent_sv",          # rep_vc_vars :143
             "replicas", "clients", "messages"]                                      # view :149-150
AUX_VARS = ["aux_svc", "aux_restart", "aux_client_acked"]                            # :145


class EvalError(Exception):
    """A TLC evaluation error (e.g. VSR.tla:421 `m.commit`)."""


class Model:
    def __init__(self, R=3, C=1, values=("v1", "v2"), L=2, restart_limit=0, assume_commit_number=False):
        self.R, self.C, self.Values, self.L = R, C, tuple(values), L
        self.RestartEmptyLimit = restart_limit
        self.assume_commit_number = assume_commit_number
        if restart_limit != 0:
            raise ValueError("RestartEmptyLimit > 0 not supported (recovery actions dead in all BASELINE configs)")


# ---- records -----------------------------------------------------------------------------------------
def rec(**kw):
    return tuple(sorted(kw.items()))


def get(r, f):
    for k, v in r:
        if k == f:
            return v
    raise EvalError("record has no field %r" % f)


def with_(r, **kw):
    d = dict(r)
    d.update(kw)
    return tuple(sorted(d.items()))


def tset(t, i, v):      # [t EXCEPT ![i] = v] for 1-based tuples
    return t[:i - 1] + (v,) + t[i:]


# ---- bag algebra (VSR.tla:228-275); a bag is a dict message -> count ---------------------------------------
def SendFunc(m, msgs):                                   # :228-231
    out = dict(msgs)
    out[m] = out[m] + 1 if m in out else 1
    return out


def BroadcastFunc(M, msg, source, msgs):                 # :233-240
    bcast = {with_(msg, dest=r) for r in range(1, M.R + 1) if r != source}
    out = {m: (c + 1 if m in bcast else c) for m, c in msgs.items()}
    for m in bcast:
        if m not in msgs:
            out[m] = 1
    return out


def DiscardFunc(m, msgs):                                # :244-245
    out = dict(msgs)
    out[m] = out[m] - 1
    return out


def ReceivableMsg(s, m, typ, r):                         # :272-275
    return get(m, "type") == typ and get(m, "dest") == r and s["messages"][m] > 0


# ---- helpers (VSR.tla:281-308) ---------------------------------------------------------------------------
def View(s, r):
    return s["rep_view_number"][r - 1]


def Primary(M, v):
    return 1 + ((v - 1) % M.R)


def IsPrimary(M, s, r):
    return Primary(M, View(s, r)) == r


def NewSVCMessage(r, view_number):
    return rec(type=StartViewChangeMsg, view_number=view_number, dest=Nil, source=r)


EmptyClientTableRow = rec(request_number=0, op_number=0, executed=True)   # :318-321


def Init(M):                                             # :323-348
    R, C = M.R, M.C
    return {
        "replicas": frozenset(range(1, R + 1)),
        "rep_status": (Normal,) * R,
        "rep_log": ((),) * R,
        "rep_view_number": (1,) * R,
        "rep_op_number": (0,) * R,
        "rep_commit_number": (0,) * R,
        "rep_peer_op_number": ((0,) * R,) * R,
        "rep_client_table": ((EmptyClientTableRow,) * C,) * R,
        "rep_svc_recv": (frozenset(),) * R,
        "rep_dvc_recv": (frozenset(),) * R,
        "rep_sent_dvc": (False,) * R,
        "rep_sent_sv": (False,) * R,
        "rep_last_normal_view": (0,) * R,
        "rep_rec_recv": (frozenset(),) * R,
        "rep_rec_number": (0,) * R,
        "clients": frozenset(range(1, C + 1)),
        "messages": {},
        "aux_svc": 0,
        "aux_restart": 0,
        "aux_client_acked": {},
    }


def upd(s, **kw):
    t = dict(s)
    t.update(kw)
    return t


def exc(s, var, r, val):
    return tset(s[var], r, val)


def msgs_sorted(s):
    """DOMAIN messages in a fixed (arbitrary) order."""
    return sorted(s["messages"].keys(), key=canon)


# ---- actions ------------------------------------------------------------------------------------------
def ReceiveClientRequest(M, s):                          # :366-394
    for r in range(1, M.R + 1):
        for c in range(1, M.C + 1):
            for v in M.Values:
                if not IsPrimary(M, s, r):
                    continue
                if s["rep_status"][r - 1] != Normal:
                    continue
                if v in s["aux_client_acked"]:
                    continue
                row = s["rep_client_table"][r - 1][c - 1]
                if get(row, "executed") is not True:
                    continue
                req_number = get(row, "request_number") + 1
                op_number = len(s["rep_log"][r - 1]) + 1
                log_entry = rec(view_number=View(s, r), operation=v, client_id=c, request_number=req_number)
                acked = dict(s["aux_client_acked"])
                acked[v] = False
                yield upd(
                    s,
                    rep_log=exc(s, "rep_log", r, s["rep_log"][r - 1] + (log_entry,)),
                    rep_op_number=exc(s, "rep_op_number", r, op_number),
                    rep_client_table=exc(s, "rep_client_table", r, tset(
                        s["rep_client_table"][r - 1], c,
                        rec(request_number=req_number, op_number=op_number, executed=False))),
                    messages=BroadcastFunc(M, rec(type=PrepareMsg, view_number=View(s, r), message=log_entry,
                                                  op_number=op_number, commit_number=s["rep_commit_number"][r - 1],
                                                  dest=Nil, source=r), r, s["messages"]),
                    aux_client_acked=acked)


def ReceivePrepareMsg(M, s):                             # :405-428
    for r in range(1, M.R + 1):
        for m in msgs_sorted(s):
            if not ReceivableMsg(s, m, PrepareMsg, r):
                continue
            if s["rep_status"][r - 1] != Normal:
                continue
            if get(m, "view_number") != View(s, r):
                continue
            if get(m, "op_number") != s["rep_op_number"][r - 1] + 1:
                continue
            entry = get(m, "message")
            table = []
            for c in range(1, M.C + 1):
                if c == get(entry, "client_id"):
                    table.append(rec(request_number=get(entry, "request_number"), op_number=get(m, "op_number"),
                                     executed=get(m, "op_number") <= get(m, "commit_number")))
                else:
                    old = s["rep_client_table"][r - 1][c - 1]
                    if M.assume_commit_number:
                        table.append(with_(old, executed=get(old, "op_number") <= get(m, "commit_number")))
                    else:
                        get(m, "commit")      # VSR.tla:421 -> EvalError: no such field
            ok = rec(type=PrepareOkMsg, view_number=View(s, r), op_number=get(m, "op_number"),
                     dest=get(m, "source"), source=r)
            yield upd(
                s,
                rep_log=exc(s, "rep_log", r, s["rep_log"][r - 1] + (entry,)),
                rep_op_number=exc(s, "rep_op_number", r, get(m, "op_number")),
                rep_commit_number=exc(s, "rep_commit_number", r, get(m, "commit_number")),
                rep_client_table=exc(s, "rep_client_table", r, tuple(table)),
                messages=SendFunc(ok, DiscardFunc(m, s["messages"])))


def ReceivePrepareOkMsg(M, s):                           # :437-447
    for r in range(1, M.R + 1):
        for m in msgs_sorted(s):
            if not ReceivableMsg(s, m, PrepareOkMsg, r):
                continue
            if not IsPrimary(M, s, r) or s["rep_status"][r - 1] != Normal:
                continue
            if get(m, "view_number") != View(s, r):
                continue
            src = get(m, "source")
            if not get(m, "op_number") > s["rep_peer_op_number"][r - 1][src - 1]:
                continue
            yield upd(
                s,
                rep_peer_op_number=exc(s, "rep_peer_op_number", r,
                                       tset(s["rep_peer_op_number"][r - 1], src, get(m, "op_number"))),
                messages=DiscardFunc(m, s["messages"]))


def IsCommitted(M, s, r, op_number):                     # :457-460
    return sum(1 for peer in range(1, M.R + 1) if s["rep_peer_op_number"][r - 1][peer - 1] >= op_number) >= M.R // 2


def ExecuteOp(M, s):                                     # :462-476
    for r in range(1, M.R + 1):
        if not IsPrimary(M, s, r) or s["rep_status"][r - 1] != Normal:
            continue
        if not s["rep_commit_number"][r - 1] < s["rep_op_number"][r - 1]:
            continue
        if not IsCommitted(M, s, r, s["rep_commit_number"][r - 1] + 1):
            continue
        op_number = s["rep_commit_number"][r - 1] + 1
        op = s["rep_log"][r - 1][op_number - 1]
        cid = get(op, "client_id")
        acked = dict(s["aux_client_acked"])
        acked[get(op, "operation")] = True
        yield upd(
            s,
            rep_commit_number=exc(s, "rep_commit_number", r, op_number),
            rep_client_table=exc(s, "rep_client_table", r, tset(
                s["rep_client_table"][r - 1], cid, with_(s["rep_client_table"][r - 1][cid - 1], executed=True))),
            aux_client_acked=acked)


def SendGetState(M, s):                                  # :496-516
    for r in range(1, M.R + 1):
        for rDest in range(1, M.R + 1):
            for m in msgs_sorted(s):
                if IsPrimary(M, s, r) or r == rDest:
                    continue
                if not ReceivableMsg(s, m, PrepareMsg, r):
                    continue
                if s["rep_status"][r - 1] != Normal:
                    continue
                if not get(m, "view_number") > View(s, r):
                    continue
                if not get(m, "op_number") > s["rep_op_number"][r - 1] + 1:
                    continue
                truncate_to = min(s["rep_commit_number"][r - 1], len(s["rep_log"][r - 1]))
                gs = rec(type=GetStateMsg, view_number=get(m, "view_number"), op_number=truncate_to,
                         dest=rDest, source=r)
                if gs in s["messages"]:                  # SendOnce :250-252
                    continue
                yield upd(
                    s,
                    rep_log=exc(s, "rep_log", r, s["rep_log"][r - 1][:truncate_to]),
                    rep_op_number=exc(s, "rep_op_number", r, truncate_to),
                    rep_view_number=exc(s, "rep_view_number", r, get(m, "view_number")),
                    rep_last_normal_view=exc(s, "rep_last_normal_view", r, get(m, "view_number")),
                    messages=SendFunc(gs, s["messages"]))


def ReceiveGetState(M, s):                               # :526-543
    for r in range(1, M.R + 1):
        for m in msgs_sorted(s):
            if not ReceivableMsg(s, m, GetStateMsg, r):
                continue
            if View(s, r) != get(m, "view_number") or s["rep_status"][r - 1] != Normal:
                continue
            if not s["rep_op_number"][r - 1] > get(m, "op_number"):
                continue
            lo, hi = get(m, "op_number") + 1, s["rep_op_number"][r - 1]
            ns = rec(type=NewStateMsg, view_number=View(s, r),
                     log=tuple((on, s["rep_log"][r - 1][on - 1]) for on in range(lo, hi + 1)),
                     first_op=lo, op_number=hi, commit_number=s["rep_commit_number"][r - 1],
                     dest=get(m, "source"), source=r)
            yield upd(s, messages=SendFunc(ns, DiscardFunc(m, s["messages"])))


def ReceiveNewState(M, s):                               # :551-567
    for r in range(1, M.R + 1):
        for m in msgs_sorted(s):
            if not ReceivableMsg(s, m, NewStateMsg, r):
                continue
            if View(s, r) != get(m, "view_number") or s["rep_status"][r - 1] != Normal:
                continue
            if s["rep_op_number"][r - 1] != get(m, "first_op") - 1:
                continue
            mlog = dict(get(m, "log"))
            new_log = tuple(s["rep_log"][r - 1][on - 1] if on <= s["rep_op_number"][r - 1] else mlog[on]
                            for on in range(1, get(m, "op_number") + 1))
            yield upd(
                s,
                rep_log=exc(s, "rep_log", r, new_log),
                rep_op_number=exc(s, "rep_op_number", r, get(m, "op_number")),
                messages=DiscardFunc(m, s["messages"]))


def TimerSendSVC(M, s):                                  # :578-590
    if not s["aux_svc"] < M.L:
        return
    for r in range(1, M.R + 1):
        if IsPrimary(M, s, r):
            continue
        yield upd(
            s,
            rep_view_number=exc(s, "rep_view_number", r, View(s, r) + 1),
            rep_status=exc(s, "rep_status", r, ViewChange),
            rep_svc_recv=exc(s, "rep_svc_recv", r, frozenset()),
            rep_dvc_recv=exc(s, "rep_dvc_recv", r, frozenset()),
            rep_sent_dvc=exc(s, "rep_sent_dvc", r, False),
            rep_sent_sv=exc(s, "rep_sent_sv", r, False),
            aux_svc=s["aux_svc"] + 1,
            messages=BroadcastFunc(M, NewSVCMessage(r, View(s, r) + 1), r, s["messages"]))


def ReceiveHigherSVC(M, s):                              # :602-613
    for m in msgs_sorted(s):
        for r in range(1, M.R + 1):
            if not ReceivableMsg(s, m, StartViewChangeMsg, r):
                continue
            if not get(m, "view_number") > View(s, r):
                continue
            yield upd(
                s,
                rep_view_number=exc(s, "rep_view_number", r, get(m, "view_number")),
                rep_status=exc(s, "rep_status", r, ViewChange),
                rep_svc_recv=exc(s, "rep_svc_recv", r, frozenset([m])),
                rep_dvc_recv=exc(s, "rep_dvc_recv", r, frozenset()),
                rep_sent_dvc=exc(s, "rep_sent_dvc", r, False),
                rep_sent_sv=exc(s, "rep_sent_sv", r, False),
                messages=BroadcastFunc(M, NewSVCMessage(r, get(m, "view_number")), r,
                                       DiscardFunc(m, s["messages"])))


def ReceiveMatchingSVC(M, s):                            # :625-634
    for m in msgs_sorted(s):
        for r in range(1, M.R + 1):
            if not ReceivableMsg(s, m, StartViewChangeMsg, r):
                continue
            if get(m, "view_number") != View(s, r) or s["rep_status"][r - 1] != ViewChange:
                continue
            yield upd(
                s,
                rep_svc_recv=exc(s, "rep_svc_recv", r, s["rep_svc_recv"][r - 1] | {m}),
                messages=DiscardFunc(m, s["messages"]))


def SendDVC(M, s):                                       # :648-669
    for r in range(1, M.R + 1):
        if s["rep_status"][r - 1] != ViewChange or s["rep_sent_dvc"][r - 1] is not False:
            continue
        if not len(s["rep_svc_recv"][r - 1]) >= M.R // 2:
            continue
        msg = rec(type=DoViewChangeMsg, view_number=View(s, r), log=s["rep_log"][r - 1],
                  last_normal_vn=s["rep_last_normal_view"][r - 1], op_number=s["rep_op_number"][r - 1],
                  commit_number=s["rep_commit_number"][r - 1], dest=Primary(M, View(s, r)), source=r)
        t = upd(s, rep_sent_dvc=exc(s, "rep_sent_dvc", r, True))
        if Primary(M, View(s, r)) == r:
            t["rep_dvc_recv"] = exc(s, "rep_dvc_recv", r, s["rep_dvc_recv"][r - 1] | {msg})
        else:
            t["messages"] = SendFunc(msg, s["messages"])
        yield t


def ReceiveHigherDVC(M, s):                              # :677-688
    for m in msgs_sorted(s):
        for r in range(1, M.R + 1):
            if not ReceivableMsg(s, m, DoViewChangeMsg, r):
                continue
            if not get(m, "view_number") > View(s, r):
                continue
            yield upd(
                s,
                rep_view_number=exc(s, "rep_view_number", r, get(m, "view_number")),
                rep_status=exc(s, "rep_status", r, ViewChange),
                rep_svc_recv=exc(s, "rep_svc_recv", r, frozenset()),
                rep_dvc_recv=exc(s, "rep_dvc_recv", r, frozenset([m])),
                rep_sent_dvc=exc(s, "rep_sent_dvc", r, False),
                rep_sent_sv=exc(s, "rep_sent_sv", r, False),
                messages=BroadcastFunc(M, NewSVCMessage(r, get(m, "view_number")), r,
                                       DiscardFunc(m, s["messages"])))


def ReceiveMatchingDVC(M, s):                            # :696-703
    for m in msgs_sorted(s):
        for r in range(1, M.R + 1):
            if not ReceivableMsg(s, m, DoViewChangeMsg, r):
                continue
            if View(s, r) != get(m, "view_number"):
                continue
            yield upd(
                s,
                rep_dvc_recv=exc(s, "rep_dvc_recv", r, s["rep_dvc_recv"][r - 1] | {m}),
                messages=DiscardFunc(m, s["messages"]))


def tlc_dvc_key(m):
    """TLC's record order on DVCs (normal-form field order, SURVEY App. B4 / trace:566)."""
    return (get(m, "view_number"), get(m, "op_number"), get(m, "commit_number"), get(m, "dest"),
            get(m, "source"), canon(get(m, "log")), get(m, "last_normal_vn"))


def HighestLog(s, r):                                    # :716-722
    dvcs = sorted(s["rep_dvc_recv"][r - 1], key=tlc_dvc_key)
    for m in dvcs:       # CHOOSE = first in TLC's set order satisfying the predicate
        if not any(get(m1, "last_normal_vn") > get(m, "last_normal_vn") or
                   (get(m1, "last_normal_vn") == get(m, "last_normal_vn") and get(m1, "op_number") > get(m, "op_number"))
                   for m1 in dvcs):
            return get(m, "log")
    raise EvalError("CHOOSE over empty set")


def SendSV(M, s):                                        # :735-760
    for r in range(1, M.R + 1):
        if s["rep_status"][r - 1] != ViewChange or s["rep_sent_sv"][r - 1] is not False:
            continue
        if not len(s["rep_dvc_recv"][r - 1]) >= M.R // 2 + 1:
            continue
        new_log = HighestLog(s, r)
        new_on = len(new_log)
        new_cn = max(get(m, "commit_number") for m in s["rep_dvc_recv"][r - 1])
        yield upd(
            s,
            rep_status=exc(s, "rep_status", r, Normal),
            rep_log=exc(s, "rep_log", r, new_log),
            rep_op_number=exc(s, "rep_op_number", r, new_on),
            rep_peer_op_number=exc(s, "rep_peer_op_number", r, (0,) * M.R),
            rep_commit_number=exc(s, "rep_commit_number", r, new_cn),
            rep_sent_sv=exc(s, "rep_sent_sv", r, True),
            rep_last_normal_view=exc(s, "rep_last_normal_view", r, View(s, r)),
            messages=BroadcastFunc(M, rec(type=StartViewMsg, view_number=View(s, r), log=new_log, op_number=new_on,
                                          commit_number=new_cn, dest=Nil, source=r), r, s["messages"]))


def ReceiveSV(M, s):                                     # :773-793
    for m in msgs_sorted(s):
        for r in range(1, M.R + 1):
            if not ReceivableMsg(s, m, StartViewMsg, r):
                continue
            if not get(m, "view_number") >= View(s, r):
                continue
            if s["rep_commit_number"][r - 1] < get(m, "op_number"):
                msgs = SendFunc(rec(type=PrepareOkMsg, view_number=get(m, "view_number"), op_number=get(m, "op_number"),
                                    dest=Primary(M, get(m, "view_number")), source=r),
                                DiscardFunc(m, s["messages"]))
            else:
                msgs = DiscardFunc(m, s["messages"])
            yield upd(
                s,
                rep_status=exc(s, "rep_status", r, Normal),
                rep_view_number=exc(s, "rep_view_number", r, get(m, "view_number")),
                rep_log=exc(s, "rep_log", r, get(m, "log")),
                rep_op_number=exc(s, "rep_op_number", r, get(m, "op_number")),
                rep_commit_number=exc(s, "rep_commit_number", r, get(m, "commit_number")),
                rep_last_normal_view=exc(s, "rep_last_normal_view", r, get(m, "view_number")),
                rep_svc_recv=exc(s, "rep_svc_recv", r, frozenset()),
                rep_dvc_recv=exc(s, "rep_dvc_recv", r, frozenset()),
                rep_sent_dvc=exc(s, "rep_sent_dvc", r, False),
                rep_sent_sv=exc(s, "rep_sent_sv", r, False),
                messages=msgs)


NEXT = [("TimerSendSVC", TimerSendSVC), ("ReceiveHigherSVC", ReceiveHigherSVC),            # :896-913
        ("ReceiveMatchingSVC", ReceiveMatchingSVC), ("SendDVC", SendDVC),
        ("ReceiveHigherDVC", ReceiveHigherDVC), ("ReceiveMatchingDVC", ReceiveMatchingDVC),
        ("SendSV", SendSV), ("ReceiveSV", ReceiveSV), ("ReceiveClientRequest", ReceiveClientRequest),
        ("ReceivePrepareMsg", ReceivePrepareMsg), ("ReceivePrepareOkMsg", ReceivePrepareOkMsg),
        ("ExecuteOp", ExecuteOp), ("SendGetState", SendGetState), ("ReceiveGetState", ReceiveGetState),
        ("ReceiveNewState", ReceiveNewState)]


def successors(M, s):
    """[(action name, successor state)] in Next order.  Recovery actions (VSR.tla:813-894) are dead with
    RestartEmptyLimit = 0."""
    out = []
    for name, fn in NEXT:
        for t in fn(M, s):
            out.append((name, t))
    return out


# ---- invariants (VSR.tla:926-950) -------------------------------------------------------------------------
def ReplicaHasOp(s, r, v):
    return any(get(e, "operation") == v for e in s["rep_log"][r - 1])


def AcknowledgedWriteNotLost(M, s):
    return all((not acked) or any(ReplicaHasOp(s, r, v) for r in range(1, M.R + 1))
               for v, acked in s["aux_client_acked"].items())


def AcknowledgedWritesExistOnMajority(M, s):
    return all((not acked) or sum(1 for r in range(1, M.R + 1) if ReplicaHasOp(s, r, v)) >= M.R // 2 + 1
               for v, acked in s["aux_client_acked"].items())


# ---- canonical values, VIEW and SYMMETRY -------------------------------------------------------------------
def canon(x):
    """A totally ordered, hashable normal form of a value."""
    if isinstance(x, bool):
        return ("b", x)
    if isinstance(x, int):
        return ("i", x)
    if isinstance(x, str):
        return ("s", x)
    if isinstance(x, (frozenset, set)):
        return ("S",) + tuple(sorted(canon(e) for e in x))
    if isinstance(x, dict):
        return ("F",) + tuple(sorted((canon(k), canon(v)) for k, v in x.items()))
    if isinstance(x, tuple):
        return ("T",) + tuple(canon(e) for e in x)
    raise TypeError(type(x))


def permute_value(x, pi):
    """Apply a Values permutation (dict value -> value) everywhere a model value of Values occurs."""
    if isinstance(x, str):
        return pi.get(x, x)
    if isinstance(x, (bool, int)):
        return x
    if isinstance(x, frozenset):
        return frozenset(permute_value(e, pi) for e in x)
    if isinstance(x, dict):
        return {permute_value(k, pi): permute_value(v, pi) for k, v in x.items()}
    if isinstance(x, tuple):
        return tuple(permute_value(e, pi) for e in x)
    raise TypeError(type(x))


def view_of(s):
    return canon(tuple(s[v] for v in VIEW_VARS))


def canonical_view(M, s, symmetry=True):
    """min over symmValues = Permutations(Values) (VSR.tla:151) of the VIEW value (VSR.tla:149-150)."""
    if not symmetry:
        return view_of(s)
    best = None
    for p in permutations(M.Values):
        pi = dict(zip(M.Values, p))
        v = view_of({k: permute_value(s[k], pi) for k in VIEW_VARS})
        if best is None or v < best:
            best = v
    return best


def bfs(M, max_depth=10 ** 9, symmetry=True):
    """Level-synchronous BFS.  Returns (levels, violation) where levels[d] = list of states first seen at depth d+1
    and violation = (depth, state) or None.  First discovery wins (shallower level)."""
    s0 = Init(M)
    seen = {canonical_view(M, s0, symmetry)}
    levels = [[s0]]
    gen = [0]
    while len(levels) < max_depth:
        nxt = []
        g = 0
        for s in levels[-1]:
            for _name, t in successors(M, s):
                g += 1
                cv = canonical_view(M, t, symmetry)
                if cv not in seen:
                    seen.add(cv)
                    nxt.append(t)
                    if not AcknowledgedWriteNotLost(M, t):
                        levels.append(nxt)
                        gen.append(g)
                        return levels, gen, (len(levels), t)
        if not nxt:
            break
        levels.append(nxt)
        gen.append(g)
    return levels, gen, None
