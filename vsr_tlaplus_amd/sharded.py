"""Multi-GPU level loop: the seen-set sharded by high fingerprint bits, one rank per GPU (≙ TLC's MultiFPSet, across devices).

Per BFS level (phases documented in include/vsrmc.h):
    expand -> all-to-all (fp, key) candidates to their owners -> owners claim + verdict -> all-to-all verdict bytes back
           -> generators materialise the winners INTO THEIR OWN next frontier (records stay with their generator; only
              16-byte candidates and verdict bytes cross ranks) -> if the ranks' frontiers are out of balance, the surplus
              is moved in bulk (export / all-to-all / append) -> commit -> small all-reduces of the level's counters.
Strict level synchrony: the set of fingerprints per level is independent of the world size.

`Exchanger` moves variable-size buckets with torch.distributed.all_to_all_single (backend "nccl" = RCCL over xGMI on
device tensors; "gloo": the same call on host tensors, device tensors are staged through the host).  `ShardedChecker`
drives any engine that implements the phase methods; `HipShardEngine` is the product engine (HIP kernels behind the C ABI).
"""
import ctypes as C

import os
import sys

import numpy as np
import torch
import torch.distributed as dist

from . import capi
from .capi import check

U64_MAX = (1 << 64) - 1


def owner_of(fp, world):
    """Owner rank of a fingerprint — must match owner_of() in csrc/vsr_kernels.hpp."""
    return ((int(fp) >> 40) & 0xFFFFFF) % world


class Exchanger:
    """Variable-size all-to-all of per-peer buckets, plus the small collectives of the level loop."""

    def __init__(self, group=None):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.bytes_sent = 0

    def _a2a(self, out, inp, out_splits, in_splits):
        if self.backend == "gloo" and inp.is_cuda:            # gloo moves host memory: stage through the host
            o = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(o, inp.cpu(), out_splits, in_splits, group=self.group)
            out.copy_(o)
        else:
            dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.group)

    def exchange(self, send_list, err=0, recv_counts=None):
        """send_list[p] = tensor of rows for peer p (same dtype / trailing shape).  Returns (recv_list, max err over ranks,
        the contiguous peer-major tensor the recv_list entries are views of).
        recv_counts: rows expected from every peer, when the caller already knows them (answers to an earlier exchange);
        otherwise a small all-to-all of (count, error code) pairs precedes the payload.
        Every bucket travels with one padding row, so that no rank ever passes an empty tensor to the collective
        (early BFS levels move nothing between most pairs of ranks)."""
        w = self.world
        dev = send_list[0].device
        send_counts = [int(t.shape[0]) for t in send_list]
        max_err = int(err)
        if recv_counts is None:
            # every peer p gets (rows I send to p, my error code)
            meta_in = torch.tensor([v for p in range(w) for v in (send_counts[p], int(err))], dtype=torch.int64)
            meta_out = torch.empty(2 * w, dtype=torch.int64)
            if self.backend == "nccl":
                mi, mo = meta_in.to(dev), meta_out.to(dev)
                dist.all_to_all_single(mo, mi, group=self.group)
                meta_out = mo.cpu()
            else:
                dist.all_to_all_single(meta_out, meta_in, group=self.group)
            recv_counts = [int(meta_out[2 * p]) for p in range(w)]
            max_err = max(int(meta_out[2 * p + 1]) for p in range(w))
        tail = tuple(send_list[0].shape[1:])
        pad = torch.zeros((1,) + tail, dtype=send_list[0].dtype, device=dev)
        inp = torch.cat([x for t in send_list for x in (t.reshape((-1,) + tail), pad)])
        out = torch.empty((sum(recv_counts) + w,) + tail, dtype=send_list[0].dtype, device=dev)
        self._a2a(out, inp.contiguous(), [c + 1 for c in recv_counts], [c + 1 for c in send_counts])
        self.bytes_sent += (sum(send_counts) - send_counts[self.rank]) * inp.element_size() * int(np.prod(tail or (1,)))
        recv, pos = [], 0
        for p in range(w):
            recv.append(out[pos: pos + recv_counts[p]])
            pos += recv_counts[p] + 1
        cat = torch.cat(recv) if w > 1 else recv[0]
        recv2, pos = [], 0
        for p in range(w):                                      # views of the padding-free contiguous tensor
            recv2.append(cat[pos: pos + recv_counts[p]])
            pos += recv_counts[p]
        return recv2, max_err, cat

    def allreduce(self, values, op):
        """values: list of python ints (< 2^63) -> list of ints reduced over ranks."""
        t = torch.tensor(values, dtype=torch.int64)
        if self.backend == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=op, group=self.group)
        return [int(x) for x in t.cpu()]

    def allgather(self, values):
        """values: list of ints -> list (per rank) of lists."""
        t = torch.tensor(values, dtype=torch.int64)
        if self.backend == "nccl":
            t = t.cuda()
        out = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(out, t, group=self.group)
        return [[int(v) for v in o.cpu()] for o in out]

    def barrier(self):
        dist.barrier(group=self.group)


class ShardError(RuntimeError):
    pass


def balance_plan(counts, tol=1.25, min_per_rank=64):
    """Deterministic (same on every rank) list of (src, dst, k): move k states from src to dst so that every rank ends
    within `tol` of the mean.  Empty when the frontier is tiny or already balanced."""
    w = len(counts)
    total = sum(counts)
    if w < 2 or total < w * min_per_rank:
        return []
    mean = total / float(w)
    if max(counts) <= tol * mean and min(counts) >= mean / tol:
        return []
    target = [total // w + (1 if r < total % w else 0) for r in range(w)]
    surplus = [[r, counts[r] - target[r]] for r in range(w) if counts[r] > target[r]]
    deficit = [[r, target[r] - counts[r]] for r in range(w) if counts[r] < target[r]]
    plan = []
    for src in surplus:
        for dst in deficit:
            if src[1] == 0:
                break
            k = min(src[1], dst[1])
            if k > 0:
                plan.append((src[0], dst[0], k))
                src[1] -= k
                dst[1] -= k
    return plan


class ShardedChecker:
    """The level loop over an engine.  Engine protocol (all tensors int64 unless noted, on the engine's device):
        expand()                          -> (list of (n_p, 2) candidate tensors per peer, err)
        claim(cands (n,2))                -> (uint8 verdict tensor (n,), err)
        materialize(verdicts per peer)    -> err                       (winners go to the local next frontier)
        count()                           -> (valid states, index range) of the local next frontier
        export(first, n)                  -> ((words, off, fp) tensors, err): records of an index window, removed locally
        append(words, off, fp)            -> err
        commit()                          -> dict(n_new, generated, deadlocks, viol_fp, viol_mask, max_bag, ...)
        lookup(key, level, by_low_bits)   -> (fingerprint, meta) or None: one step of a trace walk through this rank's seen-set
        error_text()
        local_step()                      -> dict like commit(): one whole level on this rank alone (replicated phase)
        partition()                       -> states of the current frontier this rank keeps (its own)

    Replicated phase: an engine starts with Init on every rank.  While a level has fewer than `replicate_below` new states
    every rank explores it on its own — no collective at all, the ranks compute identical state sets — because a level of
    a few thousand states costs less than the five collectives of a sharded level.  The first level that reaches
    `replicate_below` new states is partitioned by owner and the sharded protocol takes over.  0 = sharded from Init on.
    """

    def __init__(self, engine, exchanger, balance_tol=1.25, replicate_below=0):
        self.e = engine
        self.x = exchanger
        self.rank, self.world = exchanger.rank, exchanger.world
        self.level = 1
        self.distinct = 1
        self.n_frontier = 1
        self.violation = None
        self.levels = []
        self.balance_tol = balance_tol
        self.moved = 0
        self.replicated = True
        self.replicate_below = replicate_below
        if replicate_below <= 1:
            self._end_replicated()

    def _end_replicated(self):
        self.e.partition()
        self.replicated = False

    def _step_replicated(self):
        info = self.e.local_step()
        if hasattr(self.e, "set_max_bag"):                      # replicated phase: every rank sees the same level
            self.e.set_max_bag(info.get("max_bag", 0))
        self.level += 1
        self.n_frontier = info["n_new"]
        self.distinct += info["n_new"]
        viol = info["viol_fp"] if info["viol_mask"] else None
        out = dict(level=self.level, n_new=info["n_new"], generated=info["generated"], deadlocks=info["deadlocks"],
                   pending=info["pending"], distinct=self.distinct, local=info, viol_fp=viol, replicated=True,
                   per_rank_new=[info["n_new"]] * self.world)
        if info["n_new"]:
            self.levels.append(out)
        if viol is not None and self.violation is None:
            self.violation = dict(level=self.level, fp=viol, mask=info["viol_mask"])    # every rank holds the state
        elif info["n_new"] >= self.replicate_below:
            self._end_replicated()
        return out

    def _raise_if(self, err, phase):
        if err:
            raise ShardError("level %d, phase %s: error %d on some rank (local: %s)" % (self.level + 1, phase, err,
                                                                                      self.e.error_text()))

    def _rebalance(self, err):
        """All ranks: compare the sizes of the new frontiers; move the surplus (tail of the index range) where it is missing."""
        e, x, me, w = self.e, self.x, self.rank, self.world
        valid, rng = e.count()
        err = max(err, e.sticky_error() if hasattr(e, "sticky_error") else 0)
        both = x.allgather([valid, err])
        counts = [b[0] for b in both]
        self._raise_if(max(b[1] for b in both), "materialize")
        plan = balance_plan(counts, self.balance_tol)
        if not plan:
            return 0
        empty = e.empty_streams()
        send = [empty] * w
        err = 0
        hi = rng
        for src, dst, k in plan:
            if src != me:
                continue
            width = min(hi, -(-k * rng // max(1, valid)))       # index window holding about k valid records (holes are skipped)
            streams, er = e.export(hi - width, width)
            err = max(err, er)
            hi -= width
            send[dst] = streams
        got = []
        for s in range(3):                                      # words, off, fp
            r, err, _ = x.exchange([send[p][s] for p in range(w)], err)
            got.append(r)
        self._raise_if(err, "rebalance")
        err = 0
        for p in range(w):
            if p != me and got[1][p].shape[0]:
                err = max(err, e.append(got[0][p], got[1][p], got[2][p]))
                self.moved += int(got[1][p].shape[0])
        return err

    def step(self):
        if self.replicated:
            return self._step_replicated()
        e, x, me = self.e, self.x, self.rank
        cands, err = e.expand()
        err = max(err, e.sticky_error() if hasattr(e, "sticky_error") else 0)   # e.g. a failed partition at the end of the replicated phase
        recv, gerr, cat = x.exchange(cands, err)                # 2 collectives: (count, err) pairs, then the buckets
        self._raise_if(gerr, "expand")
        verdict, err = e.claim(cat)
        back, pos = [], 0
        for p in range(self.world):
            back.append(verdict[pos: pos + recv[p].shape[0]])
            pos += recv[p].shape[0]
        # the answers: as many rows come back from p as candidates went to p — no count exchange needed
        vrecv, _, _ = x.exchange(back, recv_counts=[int(c.shape[0]) for c in cands])
        err = max(err, e.materialize(vrecv))
        err = self._rebalance(err)                              # 1 all-gather (+ the moves, when the ranks drifted apart)
        info = e.commit()
        err = max(err, e.sticky_error() if hasattr(e, "sticky_error") else 0)   # count / commit failures reach every rank here
        viol_fp = info["viol_fp"] if info["viol_mask"] else U64_MAX
        # one all-gather carries every per-level figure (a 64-bit fingerprint travels as two 32-bit halves: int64 tensors)
        rows = x.allgather([info["n_new"], info["generated"], info["deadlocks"], info["pending"], viol_fp >> 32,
                            viol_fp & 0xFFFFFFFF, err, info["viol_mask"], info.get("max_bag", 0)])
        if hasattr(e, "set_max_bag"):                           # the level's largest bag over all ranks: LDS slot size of the next level
            e.set_max_bag(max(r[8] for r in rows))
        self._raise_if(max(r[6] for r in rows), "append")
        s = [sum(r[k] for r in rows) for k in range(4)]
        gviol = min((r[4] << 32) | r[5] for r in rows)
        self.level += 1
        self.n_frontier = s[0]
        self.distinct += s[0]
        out = dict(level=self.level, n_new=s[0], generated=s[1], deadlocks=s[2], pending=s[3], distinct=self.distinct,
                   local=info, viol_fp=gviol if gviol != U64_MAX else None, per_rank_new=[r[0] for r in rows])
        if s[0]:
            self.levels.append(out)
        if gviol != U64_MAX and self.violation is None:
            mask = 0
            for r in rows:
                if ((r[4] << 32) | r[5]) == gviol:
                    mask |= r[7]
            self.violation = dict(level=self.level, fp=gviol, mask=mask)
        return out

    def run(self, max_depth=None, stop_on_violation=True):
        while True:
            if max_depth is not None and self.level >= max_depth:
                return "max-depth"
            d = self.step()
            if d["n_new"] == 0:
                return "exhausted"
            if self.violation is not None and stop_on_violation:
                return "violation"

    def probe(self):
        """One level beyond the newest one without storing it (the level that no longer fits the ranks' buffers): every rank
        expands its part of the newest level in probe mode and keeps the violating successors it cannot find in its own part
        of the seen-set; those go to their owners (one all-to-all), which drop the ones they know — a successor with the VIEW
        fingerprint of an earlier state is that state for the search, whatever its aux variables say — and the smallest
        remaining fingerprint is the violation (smallest key among its copies -> its parent).  Every rank must call this.
        -> dict(level, generated, deadlocks, viol_fp or None, viol_mask, candidates)"""
        e, x, me, w = self.e, self.x, self.rank, self.world
        L = self.level
        info, pairs, err = e.probe()
        keep = pairs
        if not self.replicated:                                 # replicated phase: every rank holds the whole seen-set
            own = np.array([owner_of(int(f), w) for f in pairs[:, 0]], dtype=np.int64) if len(pairs) else np.zeros(0, dtype=np.int64)
            dev = getattr(e, "dev", None)
            send = [torch.from_numpy(pairs[own == p].view(np.int64).copy()).reshape(-1, 2) for p in range(w)]
            if dev is not None:
                send = [t.to(dev) for t in send]
            recv, gerr, cat = x.exchange(send, err)
            self._raise_if(gerr, "probe")
            got = cat.cpu().numpy().view(np.uint64).reshape(-1, 2)
            keep = got[~e.seen_before(got[:, 0], L + 1)] if len(got) else got
        elif err:
            self._raise_if(err, "probe")
        best = min(((int(f), int(k)) for f, k in keep), default=(U64_MAX, U64_MAX))
        rows = x.allgather([info["generated"], info["deadlocks"], best[0] >> 32, best[0] & 0xFFFFFFFF, best[1] >> 32, best[1] & 0xFFFFFFFF,
                            info["viol_mask"], len(keep)])
        div = w if self.replicated else 1                       # replicated: every rank probed the same level
        gbest = min(((r[2] << 32) | r[3], (r[4] << 32) | r[5]) for r in rows)
        out = dict(level=L + 1, generated=sum(r[0] for r in rows) // div, deadlocks=sum(r[1] for r in rows) // div, viol_fp=None, viol_mask=0,
                   candidates=sum(r[7] for r in rows) // div)
        if gbest[0] != U64_MAX:
            mask = 0
            for r in rows:
                mask |= r[6]
            parent = self._agree(e.lookup((gbest[1] >> 1) & ((1 << 45) - 1), L, True), "parent")
            if parent is None:
                raise ShardError("probe: the parent of the violating successor %016x is in no shard" % gbest[0])
            out.update(viol_fp=gbest[0], viol_mask=mask)
            if self.violation is None:
                self.violation = dict(level=L + 1, fp=gbest[0], mask=mask, probed=True, parent_fp=parent[0])
        return out

    def probe_trace_fps(self):
        """fingerprints of the counter-example probe() found, Init first (collective, like trace_fps)"""
        v = self.violation
        if not v or not v.get("probed"):
            raise ShardError("no violation recorded by probe()")
        return self.trace_fps(v["level"] - 1, v["parent_fp"]) + [v["fp"]]

    def save(self, prefix):
        """Checkpoint between two levels (TLC: FPSet + StateQueue checkpoint, per worker): every rank writes its shard to
        <prefix>.rank<r>of<w>, rank 0 the loop's own state to <prefix>.json.  Collective; -> None, raises on any rank's failure.
        Two phases, so that a failure on one rank cannot leave shards of different levels under one <prefix>.json: (1) every rank
        writes its shard under a level-tagged name and the error codes are gathered — on any failure the tagged files are removed
        and the previous checkpoint stays whole; (2) every rank renames its shard into place, the renames are gathered, and only
        then rank 0 replaces <prefix>.json.  A crash inside phase 2 leaves new shards under the old json: restore() compares the
        level every engine loaded with the json's and refuses."""
        import json
        final = "%s.rank%dof%d" % (prefix, self.rank, self.world)
        tagged = "%s.L%d" % (final, self.level)
        err = self.e.save(tagged)
        worst = max(r[0] for r in self.x.allgather([err]))
        if worst:
            try:
                os.remove(tagged)
            except OSError:
                pass
            self._raise_if(worst, "checkpoint")
        try:
            os.replace(tagged, final)
            err = 0
        except OSError:
            err = 1
        if max(r[0] for r in self.x.allgather([err])):
            raise ShardError("checkpoint: a rank could not move its shard into place; %s.json still names the previous checkpoint, "
                             "whose shards may be gone" % prefix)
        if self.rank == 0:
            tmp = prefix + ".json.tmp"
            with open(tmp, "w") as f:
                json.dump(dict(world=self.world, level=self.level, distinct=self.distinct, n_frontier=self.n_frontier,
                               replicated=self.replicated, replicate_below=self.replicate_below, moved=self.moved), f)
            os.replace(tmp, prefix + ".json")
        self.x.barrier()

    @classmethod
    def restore(cls, prefix, make_engine, exchanger, balance_tol=1.25):
        """Continue the run save() wrote: make_engine(path of this rank's shard file) -> engine.  Same world size.  Every engine
        must have loaded the level <prefix>.json names (a crash inside save() can leave newer shards under an older json)."""
        import json
        with open(prefix + ".json") as f:
            d = json.load(f)
        if d["world"] != exchanger.world:
            raise ShardError("the checkpoint was written by %d ranks, this run has %d" % (d["world"], exchanger.world))
        self = cls.__new__(cls)
        self.e = make_engine("%s.rank%dof%d" % (prefix, exchanger.rank, exchanger.world))
        self.x = exchanger
        self.rank, self.world = exchanger.rank, exchanger.world
        levels = [r[0] for r in self.x.allgather([int(self.e.loaded_level())])]
        if any(l != d["level"] for l in levels):
            raise ShardError("the shards under %s hold levels %s, %s.json names level %d: not one checkpoint" % (prefix, levels, prefix, d["level"]))
        self.level, self.distinct, self.n_frontier = d["level"], d["distinct"], d["n_frontier"]
        self.violation, self.levels = None, []
        self.balance_tol, self.moved = balance_tol, d["moved"]
        self.replicated, self.replicate_below = d["replicated"], d["replicate_below"]
        return self

    def _agree(self, hit, what="state"):
        """hit = (fingerprint, meta[, matches]) on the rank(s) that found something, None elsewhere -> the one pair on every rank
        (or None).  Whole (fingerprint, meta) pairs are gathered — a per-field reduction would splice halves of different states
        when two ranks answer — and more than one answer over all ranks is an AMBIGUOUS pointer (the 45 fingerprint bits a child
        keeps of its parent match states of that level on several ranks, or several on one): reported, not guessed."""
        n = 0 if hit is None else (hit[2] if len(hit) > 2 else 1)
        a, b = (hit[0], hit[1]) if hit is not None else (0, 0)
        rows = self.x.allgather([n, a >> 32, a & 0xFFFFFFFF, b >> 32, b & 0xFFFFFFFF])
        answers = sorted({((r[1] << 32) | r[2], (r[3] << 32) | r[4]) for r in rows if r[0]})
        if not answers:
            return None
        # the states of the replicated early levels sit in every rank's table: identical answers are one answer
        if len({f for f, _ in answers}) > 1 or any(r[0] > 1 for r in rows):
            raise ShardError("trace walk: ambiguous predecessor pointer — %d different states match the 45 fingerprint bits of one %s over "
                             "the ranks (expected about once in 2^45 / level size steps)" % (max(len(answers), max(r[0] for r in rows)), what))
        return answers[0]                                       # same fingerprint: same slot content up to the `taken` bit; the smallest

    # ---- levels beyond the record buffers: the protocol of csrc/vsr_shard_loop.hpp's second half, over an engine's deep_* phases ----
    def _deep_pass(self, src, level, mode):
        """one collective pass that INSERTS a level: expand `src` (states of level - 1), announce what other ranks own, claim, verdicts back
        -> (records the pass yields on this rank, its local figures)"""
        e, x = self.e, self.x
        cands, err = e.deep_expand(src, level, mode)
        recv, gerr, cat = x.exchange(cands, err)
        self._raise_if(gerr, "deep expand")
        verdict, err = e.deep_claim(cat, level, mode)
        back, pos = [], 0
        for p in range(self.world):
            back.append(verdict[pos: pos + recv[p].shape[0]])
            pos += recv[p].shape[0]
        # the answers: as many rows come back from p as candidates went to p — no count exchange needed
        vrecv, _, _ = x.exchange(back, recv_counts=[int(c.shape[0]) for c in cands])
        self._raise_if(max(x.allreduce([int(err)], dist.ReduceOp.MAX)), "deep claim")
        return e.deep_apply(vrecv)

    def _any(self, flag):
        return bool(self.x.allreduce([1 if flag else 0], dist.ReduceOp.MAX)[0])

    def deepen(self, slice_size=64):
        """One more level beyond the ranks' record buffers (collective; the Python counterpart of vsrmc_shard_loop_deepen, for engines
        with deep_* phases — the CPU stand-in of the tests; the product runs the C++ loop).  The first call makes level L+1 a virtual
        level; every further call descends from the newest stored level: regenerates levels L+1 .. L+j-1 slice inside slice, inserts level
        L+j and probes level L+j+1.  slice_size: source states per pass (small in the tests, so that every loop runs many times and the
        ranks run out of work at different moments).  -> (inserted dict, probed dict or None), figures over all ranks."""
        e, x = self.e, self.x
        if not hasattr(self, "deep"):
            self.deep, self._deep_base = 0, None
        if self.replicated:
            raise ShardError("the replicated phase stores its levels")
        L = self.level
        acc = dict(generated=0, deadlocks=0, n_new=0, viol_fp=U64_MAX, viol_mask=0, fx=0, fs=0)
        prb = dict(generated=0, deadlocks=0, viol_mask=0)
        bad = []

        def add(st):
            for k in ("generated", "deadlocks", "n_new"):
                acc[k] += st[k]
            acc["fx"] ^= st["fx"]
            acc["fs"] = (acc["fs"] + st["fs"]) & U64_MAX
            acc["viol_mask"] |= st["viol_mask"]
            acc["viol_fp"] = min(acc["viol_fp"], st["viol_fp"])

        def slices(src):
            a = 0
            while self._any(a < len(src)):
                yield src[a: a + slice_size]
                a += slice_size

        target = L + self.deep + 1
        if self.deep == 0:
            for part in slices(e.deep_source()):
                _, st = self._deep_pass(part, L + 1, "insert")
                add(st)
        else:
            e.deep_new_descent()                                 # (nothing to clear, nobody to wait for: a regeneration is local)

            def descend(src, lv):
                for part in slices(src):
                    if lv + 1 < target:
                        # a regenerated level: the rank rebuilds the states ITS candidates inserted (its winner set), once per descent — no
                        # announcement, no exchange (rounds 3-4 asked the owners about every candidate of every regenerating pass)
                        descend(e.deep_regen(part, lv + 1), lv + 1)
                    else:
                        out, st = self._deep_pass(part, lv + 1, "normal")
                        add(st)
                        info, pairs = e.deep_probe(out, lv + 2)
                        for k in ("generated", "deadlocks"):
                            prb[k] += info[k]
                        prb["viol_mask"] |= info["viol_mask"]
                        bad.extend(pairs)
            descend(e.deep_source(), L)
        rows = x.allgather([acc["generated"], acc["deadlocks"], acc["n_new"], acc["fx"] >> 32, acc["fx"] & 0xFFFFFFFF, acc["fs"] >> 32, acc["fs"] & 0xFFFFFFFF,
                            acc["viol_fp"] >> 32, acc["viol_fp"] & 0xFFFFFFFF, acc["viol_mask"], prb["generated"], prb["deadlocks"], prb["viol_mask"]])
        fx = fs = 0
        for r in rows:
            fx ^= (r[3] << 32) | r[4]
            fs = (fs + ((r[5] << 32) | r[6])) & U64_MAX
        viol = min((r[7] << 32) | r[8] for r in rows)
        ins = dict(level=target, generated=sum(r[0] for r in rows), deadlocks=sum(r[1] for r in rows), n_new=sum(r[2] for r in rows), fp_xor=fx, fp_sum=fs,
                   viol_fp=None if viol == U64_MAX else viol, viol_mask=0)
        if ins["n_new"] == 0:
            ins["level"] = target - 1
            return ins, None
        self.deep += 1
        self.distinct += ins["n_new"]
        if viol != U64_MAX:
            for r in rows:
                ins["viol_mask"] |= r[9]
            if self.violation is None:
                self.violation = dict(level=target, fp=viol, mask=ins["viol_mask"])
            return ins, None
        if target == L + 1:
            return ins, None                                     # the first pass probes nothing
        # the probe's candidates: shown to their owners, which drop what they have seen below the probed level
        own = [[] for _ in range(self.world)]
        for fp, key in bad:
            own[owner_of(int(fp), self.world)].extend((int(fp), int(key)))
        send = [torch.from_numpy(np.array(o, dtype=np.uint64).astype(np.int64).reshape(-1, 2)) for o in own]
        _, gerr, cat = x.exchange(send, 0)
        self._raise_if(gerr, "deep probe")
        got = cat.cpu().numpy().astype(np.int64).view(np.uint64).reshape(-1, 2)
        keep = got[~e.seen_before(got[:, 0], target + 1)] if len(got) else got
        best = min(((int(f), int(k)) for f, k in keep), default=(U64_MAX, U64_MAX))
        rows2 = x.allgather([best[0] >> 32, best[0] & 0xFFFFFFFF, best[1] >> 32, best[1] & 0xFFFFFFFF])
        gbest = min(((r[0] << 32) | r[1], (r[2] << 32) | r[3]) for r in rows2)
        probed = dict(level=target + 1, generated=sum(r[10] for r in rows), deadlocks=sum(r[11] for r in rows), viol_fp=None, viol_mask=0)
        if gbest[0] != U64_MAX:
            for r in rows:
                probed["viol_mask"] |= r[12]
            parent = self._agree(e.lookup((gbest[1] >> 1) & ((1 << 45) - 1), target, True), "parent")
            if parent is None:
                raise ShardError("deep probe: the parent of the violating successor %016x is in no shard" % gbest[0])
            probed["viol_fp"] = gbest[0]
            if self.violation is None:
                self.violation = dict(level=target + 1, fp=gbest[0], mask=probed["viol_mask"], probed=True, parent_fp=parent[0])
        return ins, probed

    def trace_fps(self, level, fp):
        """Walk the predecessor pointers — they live in the seen-set slots, i.e. on the owner of each state — from the level-`level`
        state with fingerprint `fp` back to Init; every rank must call this.  -> the fingerprints of the path, Init first
        (replay_fps re-executes it).  Two small all-reduces per level.
        meta = level(9) << 55 | auxkey(9) << 46 | parent fingerprint bits(45) << 1 | taken(1)."""
        fps = [fp]
        for l in range(level, 1, -1):
            hit = self._agree(self.e.lookup(fp, l, False))
            if hit is None or (hit[1] >> 55) != l:
                raise ShardError("trace walk: no level-%d state with fingerprint %016x in any shard" % (l, fp))
            parent = self._agree(self.e.lookup((hit[1] >> 1) & ((1 << 45) - 1), l - 1, True), "parent")
            if parent is None:
                raise ShardError("trace walk: the parent of %016x is in no shard" % fp)
            fp = parent[0]
            fps.append(fp)
        return fps[::-1]


class HipShardEngine:
    """The product engine: one rank's shard of the checker, HIP kernels behind the C ABI (include/vsrmc.h)."""

    def __init__(self, model, rank, world, device=0, table_log2=26, frontier_words=1 << 27, frontier_states=1 << 22,
                 pending_entries=1 << 23, cand_cap=1 << 22, rec_cap=1 << 21, rec_words_cap=1 << 26, keep_trace=True,
                 trace_entries=0, exact_ties=False, filter_log2=0, recover=None, native_only=False):
        """cand_cap: (fp, key) candidates per peer and level; rec_cap / rec_words_cap: records / words one rebalancing move
        to one peer may carry.  recover: this rank's checkpoint file (save()), written by the same rank of the same world.
        table_log2 / frontier_words / frontier_states / pending_entries / cand_cap = 0: sized from the free memory of the device
        (vsrmc_options with zeros; cand_cap: its share of the fifth the checker leaves for the exchange buffers).
        native_only: the engine will run under the C++ level loop, which owns its exchange buffers — none are allocated here."""
        self.model, self.rank, self.world, self.device = model, rank, world, device
        if not torch.cuda.is_available():
            # libvsrmc.so and torch both bind libamdhip64.so.7 (ROCm's resp. torch's bundled copy); the first one loaded serves the
            # whole process, and torch cannot enumerate devices on ROCm's copy: `import torch` (or this module) before the first
            # vsrmc call in a process that needs both
            raise ShardError("torch sees no GPU (no device, or libvsrmc.so was loaded before torch in this process: import "
                             "vsr_tlaplus_amd.sharded / torch first)")
        o = capi.Options()
        capi.load().vsrmc_options_default(C.byref(o))
        o.device, o.table_log2 = device, table_log2
        o.frontier_words, o.frontier_states, o.pending_entries = frontier_words, frontier_states, pending_entries
        o.keep_trace, o.trace_entries, o.rank, o.world = int(keep_trace), trace_entries, rank, world
        o.exact_ties, o.filter_log2 = int(exact_ties), filter_log2
        self._h = C.c_void_p()
        if recover is None:
            check(capi.load().vsrmc_checker_create(model._h, C.byref(o), C.byref(self._h)))
        else:
            check(capi.load().vsrmc_checker_load(model._h, C.byref(o), os.fsencode(recover), C.byref(self._h)))
        dev = torch.device("cuda", device)
        self.dev = dev
        check(capi.load().vsrmc_checker_options(self._h, C.byref(o)))       # sizes left 0 were derived from the free device memory
        self.options = o
        if not cand_cap and world == 1:
            cand_cap = 1 << 16                                  # nobody to announce to
        if not cand_cap:
            # 42 bytes per candidate and peer (16 sent + 16 received + 8 beside it + 2 verdict bytes); the checker's record memory is
            # 19.7 bytes per frontier word and 4/5 of what the exchange buffers + records share: 3/4 of the remaining fifth
            cand_cap = max(1 << 16, int(0.75 * 0.25 * 19.7 * int(o.frontier_words) / (42.0 * world)))
        self.cand_cap, self.rec_cap, self.rec_words_cap = cand_cap, rec_cap, rec_words_cap
        if not native_only:
            self.cand_send = torch.zeros((world, cand_cap, 2), dtype=torch.int64, device=dev)
            self.verdict_in = torch.zeros((world, cand_cap), dtype=torch.uint8, device=dev)
            self.io = capi.ShardIO(self.cand_send.data_ptr(), cand_cap)
        self._export_bufs = []                                  # allocated on first use: rebalancing is rare
        self._cand_counts = [0] * world
        self._err_text = ""
        self.kernel_ms = dict(expand=0.0, materialize=0.0)

    def _call(self, rc):
        if rc != 0:
            self._err_text = capi.load().vsrmc_last_error().decode()
            return -rc if rc < 0 else rc
        return 0

    def error_text(self):
        return self._err_text

    def local_distinct(self):
        return self._frontier_counts()[1]

    def _frontier_counts(self):
        n = C.c_uint64()
        out = np.zeros(1, dtype=np.uint64)
        rc = capi.load().vsrmc_checker_level_fps(self._h, C.c_void_p(out.ctypes.data), 1, C.byref(n))
        return rc, int(n.value)

    def reset(self):
        check(capi.load().vsrmc_checker_reset(self._h))

    def loaded_level(self):
        """the BFS level of the engine's newest frontier (after a recovery: the level the shard file held)"""
        info = capi.LevelInfo()
        check(capi.load().vsrmc_checker_status(self._h, C.byref(info)))
        return int(info.level)

    def save(self, path):
        """this rank's shard (its part of the seen-set and of the newest level) between two levels -> error code"""
        return self._call(capi.load().vsrmc_checker_save(self._h, os.fsencode(path)))

    def probe(self):
        """the local part of the newest level expanded without storing anything -> (dict(generated, deadlocks, viol_mask),
        uint64 array (n, 2) of the violating successors' (fingerprint, key) that are not in THIS rank's seen-set, err)"""
        info = capi.LevelInfo()
        err = self._call(capi.load().vsrmc_checker_probe(self._h, C.byref(info)))
        if err:
            return dict(generated=0, deadlocks=0, viol_mask=0), np.zeros((0, 2), dtype=np.uint64), err
        n = C.c_uint64()
        err = self._call(capi.load().vsrmc_checker_probe_candidates(self._h, None, 0, C.byref(n)))
        pairs = np.zeros((max(1, n.value), 2), dtype=np.uint64)
        if not err and n.value:
            err = self._call(capi.load().vsrmc_checker_probe_candidates(self._h, C.c_void_p(pairs.ctypes.data), n.value, C.byref(n)))
        d = info.as_dict()
        self.kernel_ms["expand"] += d["expand_ms"]
        return dict(generated=d["generated"], deadlocks=d["deadlocks"], viol_mask=d["viol_mask"]), pairs[: n.value if not err else 0], err

    def seen_before(self, fps, level):
        """which of these fingerprints are states of a level < `level` in this rank's seen-set -> bool array"""
        f = np.ascontiguousarray(fps, dtype=np.uint64)
        out = np.zeros(max(1, len(f)), dtype=np.uint8)
        if len(f):
            check(capi.load().vsrmc_checker_seen_batch(self._h, C.c_void_p(f.ctypes.data), len(f), int(level), C.c_void_p(out.ctypes.data)))
        return out[: len(f)].astype(bool)

    def expand(self):
        counts = (C.c_uint64 * 8)()
        err = self._call(capi.load().vsrmc_shard_expand(self._h, C.byref(self.io), counts))
        self._cand_counts = [min(int(counts[p]), self.cand_cap) if p != self.rank else 0 for p in range(self.world)]
        if err:
            self._cand_counts = [0] * self.world
        return [self.cand_send[p, : self._cand_counts[p]] for p in range(self.world)], err

    def claim(self, cands):
        n = int(cands.shape[0])
        verdict = torch.empty(n, dtype=torch.uint8, device=self.dev)
        cands = cands.contiguous()
        torch.cuda.synchronize(self.dev)                       # the candidates were produced by a collective on torch's stream
        err = self._call(capi.load().vsrmc_shard_claim(self._h, C.c_void_p(cands.data_ptr()), n, C.c_void_p(verdict.data_ptr())))
        return verdict, err

    def materialize(self, verdicts):
        for p in range(self.world):
            if p != self.rank and self._cand_counts[p]:
                assert verdicts[p].shape[0] == self._cand_counts[p]
                self.verdict_in[p, : self._cand_counts[p]].copy_(verdicts[p])
        torch.cuda.synchronize(self.dev)
        return self._call(capi.load().vsrmc_shard_materialize(self._h, C.byref(self.io), C.c_void_p(self.verdict_in.data_ptr())))

    def count(self):
        """-> (valid states, index range); a failure is remembered and rides on the next error code this rank contributes to a
        collective (raising here would leave the other ranks blocked in their next all-gather)"""
        nv, nr = C.c_uint64(), C.c_uint64()
        self._sticky = max(getattr(self, "_sticky", 0), self._call(capi.load().vsrmc_shard_count(self._h, C.byref(nv), C.byref(nr))))
        return nv.value, nr.value

    def sticky_error(self):
        """error code of an earlier phase that could not report it itself (count, commit, set_max_bag); 0 = none"""
        return getattr(self, "_sticky", 0)

    def empty_streams(self):
        z = torch.zeros(0, dtype=torch.int64, device=self.dev)
        return (z, z, z)

    def export(self, first, n):
        words = torch.empty(self.rec_words_cap, dtype=torch.int64, device=self.dev)
        off = torch.empty(self.rec_cap, dtype=torch.int64, device=self.dev)
        fp = torch.empty(self.rec_cap, dtype=torch.int64, device=self.dev)
        torch.cuda.synchronize(self.dev)
        no, nw = C.c_uint64(), C.c_uint64()
        err = self._call(capi.load().vsrmc_shard_export(self._h, first, n, C.c_void_p(words.data_ptr()), self.rec_words_cap,
                                                        C.c_void_p(off.data_ptr()), C.c_void_p(fp.data_ptr()), self.rec_cap,
                                                        C.byref(no), C.byref(nw)))
        if err:
            return self.empty_streams(), err
        return (words[: nw.value], off[: no.value], fp[: no.value]), 0

    def append(self, words, off, fp):
        words, off, fp = words.contiguous(), off.contiguous(), fp.contiguous()
        torch.cuda.synchronize(self.dev)
        return self._call(capi.load().vsrmc_shard_append(self._h, C.c_void_p(words.data_ptr()), int(words.shape[0]),
                                                         C.c_void_p(off.data_ptr()), C.c_void_p(fp.data_ptr()), int(off.shape[0])))

    def commit(self):
        info = capi.LevelInfo()
        self._sticky = max(getattr(self, "_sticky", 0), self._call(capi.load().vsrmc_shard_commit(self._h, C.byref(info))))
        d = info.as_dict()
        self.kernel_ms["expand"] += d["expand_ms"]
        self.kernel_ms["materialize"] += d["materialize_ms"]
        self.last = d
        return d

    def local_step(self):
        info = capi.LevelInfo()
        rc = capi.load().vsrmc_shard_local_step(self._h, C.byref(info))
        if rc != 0:
            self._err_text = capi.load().vsrmc_last_error().decode()
            raise ShardError("replicated level: %s" % self._err_text)
        d = info.as_dict()
        self.kernel_ms["expand"] += d["expand_ms"]
        self.kernel_ms["materialize"] += d["materialize_ms"]
        self.last = d
        return d

    def partition(self):
        n = C.c_uint64()
        self._sticky = max(getattr(self, "_sticky", 0), self._call(capi.load().vsrmc_shard_partition(self._h, C.byref(n))))
        return n.value

    def set_max_bag(self, max_bag):
        self._sticky = max(getattr(self, "_sticky", 0), self._call(capi.load().vsrmc_shard_set_max_bag(self._h, int(max_bag))))

    def find_fp(self, fp):
        idx = C.c_uint64()
        check(capi.load().vsrmc_checker_find_fp(self._h, fp, C.byref(idx)))
        return None if idx.value == U64_MAX else idx.value

    def lookup(self, key, level, by_low_bits):
        found, fp, meta = C.c_int32(), C.c_uint64(), C.c_uint64()
        check(capi.load().vsrmc_checker_lookup(self._h, int(key), int(level), int(bool(by_low_bits)), C.byref(found), C.byref(fp),
                                               C.byref(meta)))
        return (fp.value, meta.value, found.value) if found.value else None   # found = matching states of this shard (by low bits: > 1 = ambiguous)

    def level_fps(self):
        rc, n = self._frontier_counts()
        out = np.zeros(max(1, n), dtype=np.uint64)
        nn = C.c_uint64()
        check(capi.load().vsrmc_checker_level_fps(self._h, C.c_void_p(out.ctypes.data), len(out), C.byref(nn)))
        return out[: nn.value].copy()

    def close(self):
        if self._h:
            capi.load().vsrmc_checker_destroy(self._h)
            self._h = None


def replay_fps(model, fps, device=0):
    """TLCTrace.getTrace, forward half: [(action name, wire record)] for a path given by the fingerprints of its states."""
    from .checker import ACTION_NAMES
    n = len(fps)
    cap_w = (n + 2) * int(model.layout.max_record_words)
    words = np.zeros(cap_w, dtype=np.uint64)
    off = np.zeros(n + 3, dtype=np.uint64)
    acts = np.zeros(n + 3, dtype=np.int32)
    f = np.array(list(fps), dtype=np.uint64)
    ns = C.c_uint64()
    check(capi.load().vsrmc_model_replay_fps(model._h, device, C.c_void_p(f.ctypes.data), n, C.c_void_p(words.ctypes.data), cap_w,
                                             C.c_void_p(off.ctypes.data), C.c_void_p(acts.ctypes.data), len(off), C.byref(ns)))
    return [(ACTION_NAMES[acts[t]], words[int(off[t]): int(off[t + 1])].copy()) for t in range(ns.value)]


def replay(model, ords, device=0):
    """[(action name, wire record)] for a path of ordinals from Init (simulation walks)."""
    from .checker import ACTION_NAMES
    n = len(ords)
    lay = model.layout
    cap_w = (n + 2) * int(lay.max_record_words)
    words = np.zeros(cap_w, dtype=np.uint64)
    off = np.zeros(n + 3, dtype=np.uint64)
    acts = np.zeros(n + 3, dtype=np.int32)
    o = np.array(list(ords) + [0], dtype=np.uint32)
    ns = C.c_uint64()
    check(capi.load().vsrmc_model_replay(model._h, device, C.c_void_p(o.ctypes.data), n, C.c_void_p(words.ctypes.data), cap_w,
                                         C.c_void_p(off.ctypes.data), C.c_void_p(acts.ctypes.data), len(off), C.byref(ns)))
    return [(ACTION_NAMES[acts[t]], words[int(off[t]): int(off[t + 1])].copy()) for t in range(ns.value)]

# ---------------------------------------------------------------------------------------------------------------------------
# The native level loop (csrc/vsr_shard_loop.hpp): the same protocol as ShardedChecker.step, sequenced in C++ with RCCL called
# directly (grouped ncclSend / ncclRecv), or — for runs with every rank on one GPU — over torch.distributed / gloo through two callbacks.
# ---------------------------------------------------------------------------------------------------------------------------
class RcclComm:
    """vsrmc_comm over RCCL.  The 128-byte communicator id is made by rank 0 and handed round with torch.distributed (any backend)."""

    def __init__(self, device, group=None):
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        ident = [None]
        if self.rank == 0:
            buf = (C.c_uint8 * 128)()
            check(capi.load().vsrmc_comm_rccl_unique_id(buf))
            ident = [bytes(buf)]
        dist.broadcast_object_list(ident, src=0, group=group)
        buf = (C.c_uint8 * 128).from_buffer_copy(ident[0])
        self._p = C.POINTER(capi.Comm)()
        check(capi.load().vsrmc_comm_rccl_create(buf, self.rank, self.world, int(device), C.byref(self._p)))
        self.ptr = self._p

    def close(self):
        if self._p:
            capi.load().vsrmc_comm_rccl_destroy(self._p)
            self._p = None


class TorchHostComm:
    """vsrmc_comm over torch.distributed on HOST buffers (gloo): the loop stages its buckets through pinned host memory and calls
    back into Python for the two collectives.  What the one-GPU tests run the native loop on; not a production transport."""

    def __init__(self, group=None):
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)

        def a2av(ctx, send, scnt, soff, recv, rcnt, roff, eb, stream):
            try:
                w = self.world
                sc = [int(scnt[p]) * eb for p in range(w)]
                rc = [int(rcnt[p]) * eb for p in range(w)]
                st = torch.frombuffer((C.c_uint8 * max(1, sum(sc))).from_address(send), dtype=torch.uint8)[: sum(sc)] if sum(sc) else torch.empty(0, dtype=torch.uint8)
                rt = torch.empty(sum(rc), dtype=torch.uint8)
                # every bucket travels with one padding byte: gloo does not take empty splits from every peer at once
                pad = torch.zeros(1, dtype=torch.uint8)
                parts, pos = [], 0
                for p in range(w):
                    parts += [st[pos: pos + sc[p]], pad]
                    pos += sc[p]
                out = torch.empty(sum(rc) + w, dtype=torch.uint8)
                dist.all_to_all_single(out, torch.cat(parts), [x + 1 for x in rc], [x + 1 for x in sc], group=self.group)
                pos = opos = 0
                for p in range(w):
                    rt[opos: opos + rc[p]] = out[pos: pos + rc[p]]
                    pos += rc[p] + 1
                    opos += rc[p]
                if sum(rc):
                    C.memmove(recv, rt.numpy().ctypes.data, sum(rc))
                return 0
            except Exception as e:                              # an exception must not cross the C boundary
                print("TorchHostComm.alltoallv: %s" % e, file=sys.stderr)
                return 1

        def ag(ctx, send, recv, nbytes):
            try:
                t = torch.frombuffer((C.c_uint8 * nbytes).from_address(send), dtype=torch.uint8).clone()
                out = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(self.world)]
                dist.all_gather(out, t, group=self.group)
                flat = torch.cat(out)                             # (a named tensor: the bytes must outlive the expression that takes their address)
                C.memmove(recv, flat.numpy().ctypes.data, nbytes * self.world)
                return 0
            except Exception as e:
                print("TorchHostComm.allgather: %s" % e, file=sys.stderr)
                return 1

        self._a2av, self._ag = capi.A2AV_FN(a2av), capi.AG_FN(ag)          # keep the callbacks alive
        self.comm = capi.Comm(None, self.rank, self.world, 1, 0, self._a2av, self._ag)
        self.ptr = C.pointer(self.comm)

    def close(self):
        pass


class NativeShardedChecker:
    """ShardedChecker's surface (step / run / violation / trace_fps / level / distinct / replicated) over the C++ level loop.
    engine: a HipShardEngine of this rank (single-pass levels); comm: RcclComm or TorchHostComm."""

    def __init__(self, engine, comm, replicate_below=0, cand_cap=None, rec_cap=None, rec_words_cap=None):
        self.e, self.comm = engine, comm
        self.rank, self.world = comm.rank, comm.world
        self._l = C.c_void_p()
        check(capi.load().vsrmc_shard_loop_create(engine._h, comm.ptr, int(cand_cap or engine.cand_cap), int(rec_cap or engine.rec_cap),
                                                  int(rec_words_cap or engine.rec_words_cap), int(replicate_below), C.byref(self._l)))
        self.levels, self.violation = [], None
        self._sync()

    def save(self, prefix):
        """Checkpoint between two advance() calls (collective; vsrmc_shard_loop_save): every rank its shard — also of a search that has gone beyond its
        record buffers — and the loop's state, in two phases."""
        rc = capi.load().vsrmc_shard_loop_save(self._l, os.fsencode(prefix))
        if rc != 0:
            raise ShardError("checkpoint: %s" % capi.load().vsrmc_last_error().decode())

    @classmethod
    def restore(cls, prefix, make_engine, comm, cand_cap=None, rec_cap=None, rec_words_cap=None):
        """Continue the run save() wrote (collective; same number of ranks): make_engine(path of this rank's shard file) -> HipShardEngine."""
        self = cls.__new__(cls)
        self.comm = comm
        self.rank, self.world = comm.rank, comm.world
        self.e = make_engine("%s.rank%dof%d" % (prefix, comm.rank, comm.world))
        self._l = C.c_void_p()
        rc = capi.load().vsrmc_shard_loop_restore(self.e._h, comm.ptr, int(cand_cap or self.e.cand_cap), int(rec_cap or self.e.rec_cap),
                                                  int(rec_words_cap or self.e.rec_words_cap), os.fsencode(prefix), C.byref(self._l))
        if rc != 0:
            raise ShardError("recover: %s" % capi.load().vsrmc_last_error().decode())
        self.levels, self.violation = [], None
        self._sync()
        info = capi.LevelInfo()
        check(capi.load().vsrmc_checker_status(self.e._h, C.byref(info)))
        self.depth = self.level + int(info.reserved0)           # levels beyond the newest stored one that live in the seen-sets only
        return self

    def _sync(self):
        lv, rep, vm, vl = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        d, nf, vf, mv, bs = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        check(capi.load().vsrmc_shard_loop_status(self._l, C.byref(lv), C.byref(d), C.byref(nf), C.byref(rep), C.byref(vf), C.byref(vm),
                                                  C.byref(vl), C.byref(mv), C.byref(bs)))
        self.level, self.distinct, self.n_frontier, self.replicated = lv.value, d.value, nf.value, bool(rep.value)
        self.moved, self.bytes_sent = mv.value, bs.value
        if vm.value and self.violation is None:
            self.violation = dict(level=vl.value, fp=vf.value, mask=vm.value)

    def step(self):
        g, loc = capi.LevelInfo(), capi.LevelInfo()
        rc = capi.load().vsrmc_shard_loop_step(self._l, C.byref(g), C.byref(loc))
        if rc != 0:
            raise ShardError("level %d: %s" % (self.level + 1, capi.load().vsrmc_last_error().decode()))
        was_replicated = self.replicated
        self._sync()
        gd, ld = g.as_dict(), loc.as_dict()
        if hasattr(self.e, "kernel_ms"):
            self.e.kernel_ms["expand"] += ld["expand_ms"]
            self.e.kernel_ms["materialize"] += ld["materialize_ms"]
        out = dict(level=self.level, n_new=gd["n_new"], generated=gd["generated"], deadlocks=gd["deadlocks"], pending=gd["pending"],
                   distinct=self.distinct, local=ld, viol_fp=gd["viol_fp"] if gd["viol_mask"] else None, replicated=was_replicated)
        if gd["n_new"]:
            self.levels.append(out)
        return out

    def _global_dict(self, info, local=None):
        gd = info.as_dict()
        out = dict(level=gd["level"], n_new=gd["n_new"], generated=gd["generated"], deadlocks=gd["deadlocks"], pending=gd["pending"],
                   distinct=self.distinct, viol_fp=gd["viol_fp"] if gd["viol_mask"] else None, viol_mask=gd["viol_mask"], max_bag=gd["max_bag"],
                   fp_xor=gd["fp_xor"], fp_sum=gd["fp_sum"], act_generated=gd["act_generated"], expand_ms=gd["expand_ms"],
                   materialize_ms=gd["materialize_ms"], launches=gd["pending"], record_words=gd["record_words"], seconds=gd["seconds"])
        if local is not None:
            out["local"] = local
        return out

    def advance(self):
        """One unit of progress of the automatic level scheme (vsrmc_shard_loop_advance, collective): an ordinary sharded level while
        every rank predicts that its part of the next one fits, else one pass of the deep search (the next level inserted into the
        ranks' seen-sets only, the one after it probed).  -> ("level", dict, None) | ("deep", inserted dict, probed dict or None);
        figures over all ranks."""
        a, b = capi.LevelInfo(), capi.LevelInfo()
        what = C.c_int32()
        rc = capi.load().vsrmc_shard_loop_advance(self._l, C.byref(a), C.byref(b), C.byref(what))
        if rc != 0:
            raise ShardError("level %d: %s" % (getattr(self, "depth", self.level) + 1, capi.load().vsrmc_last_error().decode()))
        was_replicated = self.replicated
        self._sync()
        if what.value == 1:
            self.depth = self.level
            out = self._global_dict(a)
            out["replicated"] = was_replicated
            out["level"] = self.level
            if out["n_new"]:
                self.levels.append(out)
            return "level", out, None
        da, db = self._global_dict(a), self._global_dict(b)
        self.depth = da["level"]
        if self.violation is not None and "probed" not in self.violation:
            self.violation["probed"] = bool(db["level"] and db["viol_mask"] and not da["viol_mask"])
        return "deep", da, (db if db["level"] else None)

    def overlap_stats(self):
        """(sharded levels that ran in slices with the exchange overlapped, slices in all)"""
        a, b = C.c_uint64(), C.c_uint64()
        check(capi.load().vsrmc_shard_loop_overlap_stats(self._l, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def room(self):
        """the seen-set shards before the next advance() (collective): 0 = room on every rank, 2 = some rank's shard is more than 85 % full"""
        st = C.c_int32()
        rc = capi.load().vsrmc_shard_loop_room(self._l, C.byref(st))
        if rc != 0:
            raise ShardError("seen-set check: %s" % capi.load().vsrmc_last_error().decode())
        if st.value == 2:
            self.room_note = capi.load().vsrmc_last_error().decode()       # which set is full on this rank, and how full
        return st.value

    def run(self, max_depth=None, stop_on_violation=True):
        self.depth = getattr(self, "depth", self.level)
        while True:
            if max_depth is not None and self.depth >= max_depth:
                return "max-depth"
            if self.room() == 2:
                return "seen-set-full"
            kind, d, p = self.advance()
            if d["n_new"] == 0:
                return "exhausted"
            if self.violation is not None and stop_on_violation:
                return "violation"

    def violation_trace_fps(self):
        """fingerprints of the counter-example of the violation the run found, Init first (collective)"""
        v = self.violation
        if v.get("probed"):
            out = np.zeros(v["level"] + 1, dtype=np.uint64)
            n = C.c_int32()
            rc = capi.load().vsrmc_shard_loop_probe_trace_fps(self._l, C.c_void_p(out.ctypes.data), len(out), C.byref(n))
            if rc != 0:
                raise ShardError("trace walk: %s" % capi.load().vsrmc_last_error().decode())
            return [int(f) for f in out[: n.value]]
        return self.trace_fps(v["level"], v["fp"])

    def trace_fps(self, level, fp):
        out = np.zeros(level, dtype=np.uint64)
        rc = capi.load().vsrmc_shard_loop_trace_fps(self._l, int(level), int(fp), C.c_void_p(out.ctypes.data))
        if rc != 0:
            raise ShardError("trace walk: %s" % capi.load().vsrmc_last_error().decode())
        return [int(f) for f in out]

    def close(self):
        if self._l:
            capi.load().vsrmc_shard_loop_destroy(self._l)
            self._l = C.c_void_p()
