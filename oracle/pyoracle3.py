"""oracle/pyoracle3.py — SECOND, INDEPENDENT CPU restatement of the third model (TEST INFRASTRUCTURE, NOT PRODUCT CODE):
/root/reference/vsr-revisited/paper/analysis/04-application-state/VR_APP_STATE.tla (cited as VRAS.tla:NNN) under its
shipped VR_APP_STATE.cfg (VIEW view, no SYMMETRY, NoProgressChangeLimit = 0).

Written directly from the TLA+ text, not from oracle/vras_oracle.cpp: states are plain Python values shaped like the TLA+
values (records = tuples of (field, value) pairs, functions = dicts / tuples, sets = frozensets), VIEW identity is value
equality of the `view` tuple (VRAS.tla:102).  Its job: pin the C++ oracle on whole small state spaces
(tests/test_model3_oracles.py).  "parity unpinned" against TLC itself (no JVM here).  Slow on purpose: small cases only.

Also holds the codec between these values and the packed record of the third model (layout: vras_oracle.cpp `encode`).
"""
Normal, ViewChange, StateTransfer = "Normal", "ViewChange", "StateTransfer"                 # VRAS.tla:51-53
PrepareMsg, PrepareOkMsg = "PrepareMsg", "PrepareOkMsg"                                     # VRAS.tla:56-62
StartViewChangeMsg, DoViewChangeMsg, StartViewMsg = "StartViewChangeMsg", "DoViewChangeMsg", "StartViewMsg"
GetStateMsg, NewStateMsg = "GetStateMsg", "NewStateMsg"
Nil, AnyDest = "Nil", "AnyDest"                                                             # VRAS.tla:64-65

ACTIONS = ["TimerSendSVC", "ReceiveHigherSVC", "ReceiveMatchingSVC", "SendDVC", "ReceiveHigherDVC", "ReceiveMatchingDVC",
           "SendSV", "ReceiveSV", "ReceiveClientRequest", "ReceivePrepareMsg", "ReceivePrepareOkMsg", "PrimaryExecuteOp",
           "SendGetState", "ReceiveGetState", "ReceiveNewState"]                            # Next, VRAS.tla:811-831

VIEW_VARS = ["rep_status", "rep_log", "rep_app_state", "rep_view_number", "rep_op_number", "rep_peer_op_number", "rep_commit_number",
             "rep_last_normal_view", "rep_rec_number", "rep_rec_recv", "rep_sent_dvc", "rep_sent_sv", "rep_recv_dvc",
             "no_progress", "no_progress_ctr", "replicas", "messages"]                      # :93-102


class EvalError(Exception):
    pass


class Model:
    def __init__(self, R=3, values=("v1", "v2"), L=2, no_progress_limit=0):
        self.R, self.Values, self.L = R, tuple(values), L
        if no_progress_limit != 0:
            raise ValueError("NoProgressChangeLimit > 0 not supported")


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


def tset(t, i, v):
    return t[:i - 1] + (v,) + t[i:]


def upd(s, **kw):
    t = dict(s)
    t.update(kw)
    return t


def exc(s, var, r, val):
    return tset(s[var], r, val)


# ---- bag algebra (VRAS.tla:170-224) -----------------------------------------------------------------------------
def SendFunc(m, msgs, deliver_count):
    out = dict(msgs)
    out[m] = out[m] + 1 if m in out else deliver_count
    return out


def BroadcastFunc(M, msg, source, msgs):
    bcast = {with_(msg, dest=r) for r in range(1, M.R + 1) if r != source}
    out = {m: (c + 1 if m in bcast else c) for m, c in msgs.items()}
    for m in bcast:
        if m not in msgs:
            out[m] = 1
    return out


def DiscardFunc(m, msgs):
    out = dict(msgs)
    out[m] = out[m] - 1
    return out


def ReceivableMsg(s, m, typ, r):                                                            # :218-223
    return (get(m, "type") == typ and (get(m, "dest") == r or (get(m, "dest") == AnyDest and get(m, "source") != r))
            and s["messages"][m] > 0)


def View(s, r):
    return s["rep_view_number"][r - 1]


def Primary(M, v):
    return 1 + ((v - 1) % M.R)


def IsNormalPrimary(M, s, r):
    return Primary(M, View(s, r)) == r and s["rep_status"][r - 1] == Normal


def IsNormalBackup(M, s, r):
    return Primary(M, View(s, r)) != r and s["rep_status"][r - 1] == Normal


def NewSVCMessage(r, dest, view_number):
    return rec(type=StartViewChangeMsg, view_number=view_number, dest=dest, source=r)


def CanProgress(s, r):
    return s["no_progress"][r - 1] is False


def LogSuffix(log, op_number):                                                               # :265-268
    """<<>> or the function op_number+1..Len(log) -> entry, kept as a tuple of (op, entry) pairs"""
    if len(log) <= op_number:
        return ()
    return tuple((op, log[op - 1]) for op in range(op_number + 1, len(log) + 1))


def MaybeExecuteOps(s, r, log, old_commit, new_commit):                                      # :277-283, AppendOps :270-275
    """-> the primed rep_app_state / rep_commit_number (log: a sequence as a tuple)"""
    if not new_commit > old_commit:
        return {}
    app = s["rep_app_state"][r - 1]
    for op in range(old_commit + 1, new_commit + 1):
        if op > len(log):
            raise EvalError("log[op] outside the log (AppendOps)")
        app = app + (log[op - 1],)
    return dict(rep_app_state=exc(s, "rep_app_state", r, app), rep_commit_number=exc(s, "rep_commit_number", r, new_commit))


def Init(M):                                                                                 # :292-315
    R = M.R
    return {
        "rep_app_state": ((),) * R, "rep_recv_dvc": (frozenset(),) * R, "rep_rec_recv": (frozenset(),) * R, "rep_rec_number": (0,) * R,
        "aux_restart": 0,
        "replicas": frozenset(range(1, R + 1)), "rep_status": (Normal,) * R, "rep_log": ((),) * R, "rep_view_number": (1,) * R,
        "rep_op_number": (0,) * R, "rep_commit_number": (0,) * R, "rep_peer_op_number": ((0,) * R,) * R,
        "rep_sent_dvc": (False,) * R, "rep_sent_sv": (False,) * R, "rep_last_normal_view": (1,) * R,
        "no_progress": (False,) * R, "no_progress_ctr": 0, "messages": {}, "aux_svc": 0, "aux_client_acked": {},
    }


def canon(x):
    if isinstance(x, dict):
        return ("D", tuple(sorted((canon(k), canon(v)) for k, v in x.items())))
    if isinstance(x, (set, frozenset)):
        return ("S", tuple(sorted(canon(e) for e in x)))
    if isinstance(x, tuple):
        return ("T", tuple(canon(e) for e in x))
    return ("A", str(type(x).__name__), x)


def msgs_sorted(s):
    return sorted(s["messages"].keys(), key=canon)


def ResetVcVars(s, r, dvcs):                                                                 # :255-258
    return dict(rep_sent_dvc=exc(s, "rep_sent_dvc", r, False), rep_sent_sv=exc(s, "rep_sent_sv", r, False),
                rep_recv_dvc=exc(s, "rep_recv_dvc", r, frozenset(dvcs)))


# ---- actions ---------------------------------------------------------------------------------------------------------
def TimerSendSVC(M, s):                                                                      # :551-565
    if not s["aux_svc"] < M.L:
        return
    for r in range(1, M.R + 1):
        if not CanProgress(s, r) or IsNormalPrimary(M, s, r):
            continue
        yield upd(s, rep_view_number=exc(s, "rep_view_number", r, View(s, r) + 1), rep_status=exc(s, "rep_status", r, ViewChange),
                  aux_svc=s["aux_svc"] + 1, messages=BroadcastFunc(M, NewSVCMessage(r, Nil, View(s, r) + 1), r, s["messages"]),
                  **ResetVcVars(s, r, ()))


def _receive_higher(M, s, typ):                                                              # :575-587 / :656-668
    for m in msgs_sorted(s):
        for r in range(1, M.R + 1):
            if not CanProgress(s, r) or not ReceivableMsg(s, m, typ, r) or not get(m, "view_number") > View(s, r):
                continue
            yield upd(s, rep_view_number=exc(s, "rep_view_number", r, get(m, "view_number")),
                      rep_status=exc(s, "rep_status", r, ViewChange),
                      messages=BroadcastFunc(M, NewSVCMessage(r, Nil, get(m, "view_number")), r, DiscardFunc(m, s["messages"])),
                      **ResetVcVars(s, r, (m,) if typ == DoViewChangeMsg else ()))       # :584 {} / :665 {m}


def ReceiveHigherSVC(M, s):
    return _receive_higher(M, s, StartViewChangeMsg)


def ReceiveHigherDVC(M, s):
    return _receive_higher(M, s, DoViewChangeMsg)


def ReceiveMatchingSVC(M, s):                                                                # :595-606
    for m in msgs_sorted(s):
        for r in range(1, M.R + 1):
            if (not CanProgress(s, r) or s["rep_status"][r - 1] != ViewChange or not ReceivableMsg(s, m, StartViewChangeMsg, r)
                    or get(m, "view_number") != View(s, r) or s["rep_sent_dvc"][r - 1] is not False):   # :602
                continue
            yield upd(s, messages=DiscardFunc(m, s["messages"]))


def ReceiveMatchingDVC(M, s):                                                                # :676-687
    for m in msgs_sorted(s):
        for r in range(1, M.R + 1):
            if (not CanProgress(s, r) or s["rep_status"][r - 1] != ViewChange or not ReceivableMsg(s, m, DoViewChangeMsg, r)
                    or get(m, "view_number") != View(s, r)):
                continue
            yield upd(s, messages=DiscardFunc(m, s["messages"]),
                      rep_recv_dvc=exc(s, "rep_recv_dvc", r, s["rep_recv_dvc"][r - 1] | {m}))    # :685


def SendDVC(M, s):                                                                           # :619-647
    for r in range(1, M.R + 1):
        if not CanProgress(s, r) or s["rep_status"][r - 1] != ViewChange or s["rep_sent_dvc"][r - 1] is not False:
            continue
        q = sum(1 for m, c in s["messages"].items()
                if get(m, "type") == StartViewChangeMsg and get(m, "dest") == r and get(m, "view_number") == View(s, r) and c == 0)
        if not q >= M.R // 2:
            continue
        msg = rec(type=DoViewChangeMsg, view_number=View(s, r), log=s["rep_log"][r - 1],
                  last_normal_vn=s["rep_last_normal_view"][r - 1], op_number=s["rep_op_number"][r - 1],
                  commit_number=s["rep_commit_number"][r - 1], dest=Primary(M, View(s, r)), source=r)
        if Primary(M, View(s, r)) == r:                                                      # :640-642 SendAsReceived, counted at once
            yield upd(s, rep_sent_dvc=exc(s, "rep_sent_dvc", r, True), messages=SendFunc(msg, s["messages"], 0),
                      rep_recv_dvc=exc(s, "rep_recv_dvc", r, s["rep_recv_dvc"][r - 1] | {msg}))
        else:                                                                                # :643-645 Send
            yield upd(s, rep_sent_dvc=exc(s, "rep_sent_dvc", r, True), messages=SendFunc(msg, s["messages"], 1))


def ValidDvc(s, r, m):                                                                       # :700-701
    return get(m, "view_number") == View(s, r)


def tlc_dvc_key(m):
    """TLC's CHOOSE takes the first satisfying element in its value order; DVC records share arity and field names and differ
    from op_number on: [view_number, type, op_number, commit_number, dest, source, log, last_normal_vn] [TLC-RECALLED]."""
    return (get(m, "op_number"), get(m, "commit_number"), get(m, "dest"), get(m, "source"))


def HighestLogMsg(s, r):                                                                     # :703-711
    cands = [m for m in s["rep_recv_dvc"][r - 1] if ValidDvc(s, r, m)]
    best = [m for m in cands if not any(
        get(m1, "last_normal_vn") > get(m, "last_normal_vn") or
        (get(m1, "last_normal_vn") == get(m, "last_normal_vn") and get(m1, "op_number") > get(m, "op_number")) for m1 in cands)]
    if not best:
        raise EvalError("CHOOSE over an empty set")
    return sorted(best, key=tlc_dvc_key)[0]


def SendSV(M, s):                                                                            # :726-754
    for r in range(1, M.R + 1):
        if not CanProgress(s, r) or s["rep_status"][r - 1] != ViewChange or s["rep_sent_sv"][r - 1] is not False:
            continue
        valid = [m for m in s["rep_recv_dvc"][r - 1] if ValidDvc(s, r, m)]
        if not len(valid) >= M.R // 2 + 1:                                                   # :732
            continue
        new_log = get(HighestLogMsg(s, r), "log")
        new_on = 0 if new_log == () else len(new_log)                                        # :713-716
        new_cn = max(get(m, "commit_number") for m in valid)                                 # :718-724
        sv = rec(type=StartViewMsg, view_number=View(s, r), log=new_log, op_number=new_on, commit_number=new_cn, dest=Nil, source=r)
        t = upd(s, rep_status=exc(s, "rep_status", r, Normal), rep_log=exc(s, "rep_log", r, new_log),
                rep_op_number=exc(s, "rep_op_number", r, new_on), rep_peer_op_number=exc(s, "rep_peer_op_number", r, (0,) * M.R),
                rep_sent_sv=exc(s, "rep_sent_sv", r, True), rep_recv_dvc=exc(s, "rep_recv_dvc", r, frozenset()),
                rep_last_normal_view=exc(s, "rep_last_normal_view", r, View(s, r)),
                messages=BroadcastFunc(M, sv, r, s["messages"]))
        t.update(MaybeExecuteOps(s, r, new_log, s["rep_commit_number"][r - 1], new_cn))      # :740
        yield t


def ReceiveSV(M, s):                                                                         # :765-788
    for m in msgs_sorted(s):
        for r in range(1, M.R + 1):
            if not CanProgress(s, r) or not ReceivableMsg(s, m, StartViewMsg, r):
                continue
            mv = get(m, "view_number")
            if not ((mv == View(s, r) and s["rep_status"][r - 1] == ViewChange) or mv > View(s, r)):
                continue
            if s["rep_commit_number"][r - 1] < get(m, "op_number"):                          # :781 (unprimed)
                ok = rec(type=PrepareOkMsg, view_number=mv, op_number=get(m, "op_number"), dest=Primary(M, mv), source=r)
                msgs = SendFunc(ok, DiscardFunc(m, s["messages"]), 1)
            else:
                msgs = DiscardFunc(m, s["messages"])
            t = upd(s, rep_status=exc(s, "rep_status", r, Normal), rep_view_number=exc(s, "rep_view_number", r, mv),
                    rep_log=exc(s, "rep_log", r, get(m, "log")), rep_op_number=exc(s, "rep_op_number", r, get(m, "op_number")),
                    rep_last_normal_view=exc(s, "rep_last_normal_view", r, mv), messages=msgs, **ResetVcVars(s, r, ()))
            t.update(MaybeExecuteOps(s, r, get(m, "log"), s["rep_commit_number"][r - 1], get(m, "commit_number")))   # :777
            yield t


def ReceiveClientRequest(M, s):                                                              # :328-349
    for r in range(1, M.R + 1):
        for v in M.Values:
            if not CanProgress(s, r) or not IsNormalPrimary(M, s, r) or v in s["aux_client_acked"]:
                continue
            op_number = len(s["rep_log"][r - 1]) + 1
            entry = rec(operation=v)
            acked = dict(s["aux_client_acked"])
            acked[v] = False
            pm = rec(type=PrepareMsg, view_number=View(s, r), message=entry, op_number=op_number,
                     commit_number=s["rep_commit_number"][r - 1], dest=Nil, source=r)
            yield upd(s, rep_log=exc(s, "rep_log", r, s["rep_log"][r - 1] + (entry,)), rep_op_number=exc(s, "rep_op_number", r, op_number),
                      messages=BroadcastFunc(M, pm, r, s["messages"]), aux_client_acked=acked)


def ReceivePrepareMsg(M, s):                                                                 # :360-380
    for r in range(1, M.R + 1):
        for m in msgs_sorted(s):
            if (not CanProgress(s, r) or not ReceivableMsg(s, m, PrepareMsg, r) or not IsNormalBackup(M, s, r)
                    or get(m, "view_number") != View(s, r) or get(m, "op_number") != s["rep_op_number"][r - 1] + 1):
                continue
            log = s["rep_log"][r - 1] + (get(m, "message"),)                                 # :369
            ok = rec(type=PrepareOkMsg, view_number=View(s, r), op_number=get(m, "op_number"), dest=get(m, "source"), source=r)
            t = upd(s, rep_log=exc(s, "rep_log", r, log), rep_op_number=exc(s, "rep_op_number", r, get(m, "op_number")),
                    messages=SendFunc(ok, DiscardFunc(m, s["messages"]), 1))
            t.update(MaybeExecuteOps(s, r, log, s["rep_commit_number"][r - 1], get(m, "commit_number")))   # :373
            yield t


def ReceivePrepareOkMsg(M, s):                                                               # :393-405
    for r in range(1, M.R + 1):
        for m in msgs_sorted(s):
            if (not CanProgress(s, r) or not IsNormalPrimary(M, s, r) or not ReceivableMsg(s, m, PrepareOkMsg, r)
                    or get(m, "view_number") != View(s, r)
                    or not get(m, "op_number") > s["rep_peer_op_number"][r - 1][get(m, "source") - 1]):
                continue
            row = tset(s["rep_peer_op_number"][r - 1], get(m, "source"), get(m, "op_number"))
            yield upd(s, rep_peer_op_number=exc(s, "rep_peer_op_number", r, row), messages=DiscardFunc(m, s["messages"]))


def PrimaryExecuteOp(M, s):                                                                  # :420-435
    for r in range(1, M.R + 1):
        if not CanProgress(s, r) or not IsNormalPrimary(M, s, r):
            continue
        cn = s["rep_commit_number"][r - 1]
        if not cn < s["rep_op_number"][r - 1]:
            continue
        if not sum(1 for p in range(M.R) if s["rep_peer_op_number"][r - 1][p] >= cn + 1) >= M.R // 2:   # IsCommitted :415-418
            continue
        if cn + 1 > len(s["rep_log"][r - 1]):
            raise EvalError("rep_log[r][new_commit] outside the log")
        op = s["rep_log"][r - 1][cn]                      # rep_log[r][cn + 1], 1-based        :429
        acked = dict(s["aux_client_acked"])
        if get(op, "operation") not in acked:
            raise EvalError("EXCEPT outside the domain")
        acked[get(op, "operation")] = True
        t = upd(s, aux_client_acked=acked)
        t.update(MaybeExecuteOps(s, r, s["rep_log"][r - 1], cn, cn + 1))                    # :431
        yield t


def SendGetState(M, s):                                                                      # :461-476
    for r in range(1, M.R + 1):
        for m in msgs_sorted(s):
            if (not CanProgress(s, r) or not IsNormalBackup(M, s, r) or not ReceivableMsg(s, m, PrepareMsg, r)
                    or not get(m, "view_number") > View(s, r) or not get(m, "op_number") > s["rep_op_number"][r - 1] + 1):
                continue
            gs = rec(type=GetStateMsg, view_number=get(m, "view_number"), op_number=s["rep_commit_number"][r - 1], dest=AnyDest, source=r)
            if gs in s["messages"]:                      # SendOnce :195-197
                continue
            yield upd(s, rep_status=exc(s, "rep_status", r, StateTransfer), messages=SendFunc(gs, s["messages"], 1))


def ReceiveGetState(M, s):                                                                   # :490-507
    for r in range(1, M.R + 1):
        for m in msgs_sorted(s):
            if (not CanProgress(s, r) or not ReceivableMsg(s, m, GetStateMsg, r) or View(s, r) != get(m, "view_number")
                    or s["rep_status"][r - 1] != Normal or not s["rep_op_number"][r - 1] > get(m, "op_number")):
                continue
            ns = rec(type=NewStateMsg, view_number=View(s, r), log=LogSuffix(s["rep_log"][r - 1], get(m, "op_number")),
                     first_op=get(m, "op_number") + 1, op_number=s["rep_op_number"][r - 1],
                     commit_number=s["rep_commit_number"][r - 1], dest=get(m, "source"), source=r)
            yield upd(s, messages=SendFunc(ns, DiscardFunc(m, s["messages"]), 1))


def ReceiveNewState(M, s):                                                                   # :516-537
    for r in range(1, M.R + 1):
        for m in msgs_sorted(s):
            if (s["rep_status"][r - 1] != StateTransfer or not CanProgress(s, r) or not ReceivableMsg(s, m, NewStateMsg, r)
                    or not get(m, "view_number") > View(s, r)):
                continue
            mlog = dict(get(m, "log"))
            old = s["rep_log"][r - 1]
            new_log = []
            for on in range(1, get(m, "op_number") + 1):                                     # :524-527
                if on < get(m, "first_op"):
                    if on > len(old):
                        raise EvalError("rep_log[r][op] outside the log")
                    new_log.append(old[on - 1])
                else:
                    if on not in mlog:
                        raise EvalError("m.log[op] outside the message log")
                    new_log.append(mlog[on])
            new_log = tuple(new_log)
            mv = get(m, "view_number")
            t = upd(s, rep_status=exc(s, "rep_status", r, Normal), rep_view_number=exc(s, "rep_view_number", r, mv),
                    rep_last_normal_view=exc(s, "rep_last_normal_view", r, mv), rep_log=exc(s, "rep_log", r, new_log),
                    rep_op_number=exc(s, "rep_op_number", r, get(m, "op_number")), messages=DiscardFunc(m, s["messages"]))
            t.update(MaybeExecuteOps(s, r, new_log, s["rep_commit_number"][r - 1], get(m, "commit_number")))   # :533
            yield t


ACTION_FUNCS = [TimerSendSVC, ReceiveHigherSVC, ReceiveMatchingSVC, SendDVC, ReceiveHigherDVC, ReceiveMatchingDVC, SendSV, ReceiveSV,
                ReceiveClientRequest, ReceivePrepareMsg, ReceivePrepareOkMsg, PrimaryExecuteOp, SendGetState, ReceiveGetState, ReceiveNewState]


def successors(M, s):
    out = []
    for name, f in zip(ACTIONS, ACTION_FUNCS):
        for t in f(M, s):
            out.append((name, t))
    return out


# ---- invariants (VRAS.tla:840-894) ---------------------------------------------------------------------------------------
def ReplicaHasOp(s, r, v):
    return any(get(e, "operation") == v for e in s["rep_log"][r - 1])


def AcknowledgedWritesExistOnMajority(M, s):
    return all(a is False or sum(1 for r in range(1, M.R + 1) if ReplicaHasOp(s, r, v)) >= M.R // 2 + 1
               for v, a in s["aux_client_acked"].items())


def NoLogDivergence(M, s):
    for on in range(1, len(M.Values) + 1):
        for r1 in range(1, M.R + 1):
            for r2 in range(1, M.R + 1):
                if on <= s["rep_commit_number"][r1 - 1] and on <= s["rep_commit_number"][r2 - 1]:
                    if s["rep_log"][r1 - 1][on - 1] != s["rep_log"][r2 - 1][on - 1]:
                        return False
    return True


def NoAppStateDivergence(M, s):                                                              # :852-858
    for on in range(1, len(M.Values) + 1):
        for r1 in range(1, M.R + 1):
            for r2 in range(1, M.R + 1):
                if on <= s["rep_commit_number"][r1 - 1] and on <= s["rep_commit_number"][r2 - 1]:
                    if (s["rep_app_state"][r1 - 1][on - 1] != s["rep_app_state"][r2 - 1][on - 1]
                            and s["rep_log"][r1 - 1][on - 1] == s["rep_app_state"][r1 - 1][on - 1]):
                        return False
    return True


def CommitNumberNeverHigherThanOpNumber(M, s):
    return all(s["rep_commit_number"][r] <= s["rep_op_number"][r] for r in range(M.R))


def invariant_mask(M, s):
    """mask of VIOLATED invariants, bits as in vras_oracle.hpp (bits 1-4 = the four of VR_APP_STATE.cfg:37-40)"""
    return ((0 if AcknowledgedWritesExistOnMajority(M, s) else 2) | (0 if NoLogDivergence(M, s) else 4) |
            (0 if CommitNumberNeverHigherThanOpNumber(M, s) else 8) | (0 if NoAppStateDivergence(M, s) else 16))


def view_of(s):
    return canon(tuple(s[v] for v in VIEW_VARS))


def bfs(M, max_depth=10 ** 9):
    """-> (levels: list of lists of states, generated per level, first violating (state, mask) or None)"""
    s0 = Init(M)
    seen = {view_of(s0)}
    levels, gen, viol = [[s0]], [0], None
    while len(levels) < max_depth and viol is None:
        nxt, g = [], 0
        for s in levels[-1]:
            for _, t in successors(M, s):
                g += 1
                k = view_of(t)
                if k not in seen:
                    seen.add(k)
                    nxt.append(t)
                    if viol is None and invariant_mask(M, t):
                        viol = (t, invariant_mask(M, t))
        if not nxt:
            break
        levels.append(nxt)
        gen.append(g)
    return levels, gen, viol


# ---- codec: python state <-> packed record (layout of vras_oracle.cpp `encode`) -------------------------------------------------
STATUS_CODE = {Normal: 0, ViewChange: 1, StateTransfer: 2}
TYPE_CODE = {StartViewChangeMsg: 1, PrepareMsg: 2, PrepareOkMsg: 3, DoViewChangeMsg: 4, StartViewMsg: 5, GetStateMsg: 6, NewStateMsg: 7}
CODE_STATUS = {v: k for k, v in STATUS_CODE.items()}
CODE_TYPE = {v: k for k, v in TYPE_CODE.items()}


def enc_msg(M, m, count):
    d = dict(m)
    t = TYPE_CODE[d["type"]]
    lg = 0
    if t == 2:
        lg = 1 | (M.Values.index(get(d["message"], "operation")) << 3)
    elif t in (4, 5):
        for i, e in enumerate(d["log"]):
            lg |= (1 | (M.Values.index(get(e, "operation")) << 3)) << (8 * i)
    elif t == 7:
        for on, e in d["log"]:
            lg |= (1 | (M.Values.index(get(e, "operation")) << 3)) << (8 * (on - 1))
    dest = 7 if d["dest"] == AnyDest else d["dest"]
    return (t | (d["view_number"] << 3) | (dest << 6) | (d["source"] << 9) | (d.get("op_number", 0) << 12) |
            (d.get("commit_number", 0) << 14) | (d.get("last_normal_vn", 0) << 16) | (d.get("first_op", 0) << 19) | (count << 21) | (lg << 32))


def dec_msg(M, w):
    t = w & 7
    view, dest, src = (w >> 3) & 7, (w >> 6) & 7, (w >> 9) & 7
    op, commit, lnv, first = (w >> 12) & 3, (w >> 14) & 3, (w >> 16) & 7, (w >> 19) & 3
    count, lg = (w >> 21) & 3, w >> 32
    ent = lambda b: rec(operation=M.Values[(b >> 3) & 3])                                   # noqa: E731
    seq = tuple(ent((lg >> (8 * i)) & 0xFF) for i in range(3) if (lg >> (8 * i)) & 7)
    typ = CODE_TYPE[t]
    dest = AnyDest if dest == 7 else dest
    if t == 1:
        m = rec(type=typ, view_number=view, dest=dest, source=src)
    elif t == 2:
        m = rec(type=typ, view_number=view, message=ent(lg & 0xFF), op_number=op, commit_number=commit, dest=dest, source=src)
    elif t in (3, 6):
        m = rec(type=typ, view_number=view, op_number=op, dest=dest, source=src)
    elif t == 4:
        m = rec(type=typ, view_number=view, log=seq, last_normal_vn=lnv, op_number=op, commit_number=commit, dest=dest, source=src)
    elif t == 5:
        m = rec(type=typ, view_number=view, log=seq, op_number=op, commit_number=commit, dest=dest, source=src)
    else:
        log = tuple((on, ent((lg >> (8 * (on - 1))) & 0xFF)) for on in range(first, op + 1))
        m = rec(type=typ, view_number=view, log=log, first_op=first, op_number=op, commit_number=commit, dest=dest, source=src)
    return m, count


def pack(M, s):
    msgs = sorted(s["messages"].items(), key=lambda kv: canon(kv[0]))
    hdr = len(msgs) | (s["aux_svc"] << 8) | (s["no_progress_ctr"] << 20)
    for i, v in enumerate(M.Values):
        if v in s["aux_client_acked"]:
            hdr |= (2 if s["aux_client_acked"][v] else 1) << (11 + 2 * i)
    words = [hdr]
    for i in range(M.R):
        A = (STATUS_CODE[s["rep_status"][i]] | (s["rep_view_number"][i] << 2) | (s["rep_op_number"][i] << 5) |
             (s["rep_commit_number"][i] << 7) | (s["rep_last_normal_view"][i] << 9) | (int(s["rep_sent_dvc"][i]) << 12) |
             (int(s["rep_sent_sv"][i]) << 13) | (int(s["no_progress"][i]) << 14))
        for p in range(M.R):
            A |= s["rep_peer_op_number"][i][p] << (15 + 2 * p)
        assert s["rep_op_number"][i] == len(s["rep_log"][i])
        for k, e in enumerate(s["rep_log"][i]):
            A |= (1 | (M.Values.index(get(e, "operation")) << 1)) << (25 + 3 * k)
        assert len(s["rep_app_state"][i]) == s["rep_commit_number"][i]
        for k, e in enumerate(s["rep_app_state"][i]):
            A |= M.Values.index(get(e, "operation")) << (34 + 2 * k)
        words.append(A)
        B = 0
        for m in s["rep_recv_dvc"][i]:
            d = dict(m)
            assert d["type"] == DoViewChangeMsg and d["dest"] == i + 1 and d["op_number"] == len(d["log"])
            assert (B & 7) in (0, d["view_number"])
            sh = 3 + 17 * (d["source"] - 1)
            assert not (B >> sh) & 1
            lg = 0
            for k, e in enumerate(d["log"]):
                lg |= (1 | (M.Values.index(get(e, "operation")) << 1)) << (3 * k)
            B |= d["view_number"] | ((1 | (d["last_normal_vn"] << 1) | (d["op_number"] << 4) | (d["commit_number"] << 6) | (lg << 8)) << sh)
        words.append(B)
    for m, c in msgs:
        words.append(enc_msg(M, m, c))
    return words


def unpack(M, words):
    hdr = int(words[0])
    nmsg = hdr & 0xFF
    R = M.R
    s = Init(M)
    acked = {}
    for i, v in enumerate(M.Values):
        a = (hdr >> (11 + 2 * i)) & 3
        if a:
            acked[v] = a == 2
    cols = {k: [] for k in ("rep_status", "rep_view_number", "rep_op_number", "rep_commit_number", "rep_last_normal_view",
                            "rep_sent_dvc", "rep_sent_sv", "no_progress", "rep_peer_op_number", "rep_log", "rep_app_state",
                            "rep_recv_dvc")}
    for i in range(R):
        A, B = int(words[1 + 2 * i]), int(words[2 + 2 * i])
        cols["rep_status"].append(CODE_STATUS[A & 3])
        cols["rep_view_number"].append((A >> 2) & 7)
        cols["rep_op_number"].append((A >> 5) & 3)
        cols["rep_commit_number"].append((A >> 7) & 3)
        cols["rep_last_normal_view"].append((A >> 9) & 7)
        cols["rep_sent_dvc"].append(bool((A >> 12) & 1))
        cols["rep_sent_sv"].append(bool((A >> 13) & 1))
        cols["no_progress"].append(bool((A >> 14) & 1))
        cols["rep_peer_op_number"].append(tuple((A >> (15 + 2 * p)) & 3 for p in range(R)))
        cols["rep_log"].append(tuple(rec(operation=M.Values[((A >> (25 + 3 * k)) >> 1) & 3]) for k in range(3) if (A >> (25 + 3 * k)) & 1))
        cols["rep_app_state"].append(tuple(rec(operation=M.Values[(A >> (34 + 2 * k)) & 3]) for k in range((A >> 7) & 3)))
        dvcs = set()
        for src in range(1, R + 1):
            slot = (B >> (3 + 17 * (src - 1))) & 0x1FFFF
            if slot & 1:
                lg = (slot >> 8) & 0x1FF
                log = tuple(rec(operation=M.Values[((lg >> (3 * k)) >> 1) & 3]) for k in range(3) if (lg >> (3 * k)) & 1)
                dvcs.add(rec(type=DoViewChangeMsg, view_number=B & 7, log=log, last_normal_vn=(slot >> 1) & 7, op_number=(slot >> 4) & 3,
                             commit_number=(slot >> 6) & 3, dest=i + 1, source=src))
        cols["rep_recv_dvc"].append(frozenset(dvcs))
    msgs = {}
    for j in range(nmsg):
        m, c = dec_msg(M, int(words[1 + 2 * R + j]))
        msgs[m] = c
    return upd(s, messages=msgs, aux_svc=(hdr >> 8) & 7, no_progress_ctr=(hdr >> 20) & 7, aux_client_acked=acked,
               **{k: tuple(v) for k, v in cols.items()})


def normalise(M, words):
    return pack(M, unpack(M, words))
