"""TEST INFRASTRUCTURE: a CPU stand-in for one rank's shard engine, built on the CPU oracle, implementing the phase protocol
of vsr_tlaplus_amd/sharded.py (expand / claim / materialize / append / commit).  It lets the world_size-2 gloo test exercise
the orchestrator + exchange protocol without a GPU.  Never imported by product code."""
import numpy as np
import torch

from oracle import orc

U64_MAX = (1 << 64) - 1


def _i64(x):
    """python ints (possibly >= 2^63) -> int64 tensor with the same bit patterns"""
    return torch.from_numpy(np.array(x, dtype=np.uint64).astype(np.int64).reshape(-1))


def _u64(t):
    return [int(v) for v in t.cpu().numpy().astype(np.int64).view(np.uint64).reshape(-1)]


class FakeShardEngine:
    def __init__(self, params, rank, world, owner_of):
        self.P, self.rank, self.world, self.owner_of = params, rank, world, owner_of
        self.seen = {}            # fp -> best key (level << 55 | auxkey << 46 | parent fingerprint bits(45) << 1)
        self.level = 1
        init = orc.init_record(params)
        fp, ak = orc.fingerprint(params, init)
        # replicated start: Init on every rank (sharded.ShardedChecker partitions when its replicated phase ends)
        self.frontier, self.fps = [init], [fp]
        self.seen[fp] = self._key(1, ak, 0, 0)
        self.total = len(self.frontier)
        self._err = ""

    def local_step(self):
        """one whole level on this rank alone: every successor is claimed locally, whoever owns it"""
        owner_of, self.owner_of = self.owner_of, (lambda fp, world: self.rank)
        try:
            self.expand()
            self.materialize([None] * self.world)
        finally:
            self.owner_of = owner_of
        return self.commit()

    def partition(self):
        keep = set(i for i, f in enumerate(self.fps) if f is not None and self.owner_of(f, self.world) == self.rank)
        # withdrawn states become holes of the index range, as in the product engine
        self.frontier = [self.frontier[i] if i in keep else None for i in range(len(self.frontier))]
        self.fps = [self.fps[i] if self.frontier[i] is not None else None for i in range(len(self.fps))]
        return len(keep)

    def _key(self, level, ak, parent_fp, ordinal):
        return (level << 55) | (ak << 46) | ((parent_fp & ((1 << 45) - 1)) << 1)

    def error_text(self):
        return self._err

    def local_distinct(self):
        return sum(1 for r in self.frontier if r is not None)

    def _claim(self, fp, key):
        """-> True if the slot belongs to this level (candidate may still win)"""
        cur = self.seen.get(fp)
        if cur is not None and (cur >> 55) < self.level + 1:
            return False
        self.seen[fp] = key if cur is None else min(cur, key)
        return True

    def expand(self):
        self.generated = self.deadlocks = 0
        self.local_pending, self.sent = [], [[] for _ in range(self.world)]
        self.succ_cache = {}
        self.where = {}           # (fp, key) -> (parent index, ordinal): the generator's own bookkeeping (the product keeps it
        for pidx, rec in enumerate(self.frontier):          # beside the candidate lists; it never crosses ranks)
            if rec is None:                                     # withdrawn by partition()
                continue
            succ = orc.successors(self.P, rec)
            self.succ_cache[pidx] = succ
            self.generated += len(succ)
            self.deadlocks += 0 if succ else 1
            for k, s in enumerate(succ):
                key = self._key(self.level + 1, s["auxkey"], self.fps[pidx], k)
                self.where.setdefault((s["fp"], key), (pidx, k))
                o = self.owner_of(s["fp"], self.world)
                if o == self.rank:
                    if self._claim(s["fp"], key):
                        self.local_pending.append((s["fp"], key))
                else:
                    self.sent[o].append((s["fp"], key))
        out = []
        for o in range(self.world):
            flat = [x for fk in self.sent[o] for x in fk]
            out.append(_i64(flat).reshape(-1, 2))
        return out, 0

    def claim(self, cands):
        vals = _u64(cands)
        self.recv = [(vals[2 * i], vals[2 * i + 1]) for i in range(len(vals) // 2)]
        alive = [self._claim(fp, key) for fp, key in self.recv]
        verdict, taken = [], set()                              # exactly-once: of equal (fp, key) candidates one is told "yes"
        for a, (fp, key) in zip(alive, self.recv):
            win = a and self.seen[fp] == key and fp not in taken
            if win:
                taken.add(fp)
            verdict.append(1 if win else 0)
        return torch.tensor(verdict, dtype=torch.uint8), 0

    def _record_of(self, fp, key):
        pidx, k = self.where[(fp, key)]
        return self.succ_cache[pidx][k]

    def materialize(self, verdicts):
        self.next_frontier, self.next_fps = [], []
        self.viol_fp, self.viol_mask, self.max_bag = U64_MAX, 0, 0
        done = set()                                            # exactly-once: candidates with equal (fp, key) are one state
        for fp, key in self.local_pending:
            if self.seen[fp] == key and (fp, key) not in done:
                done.add((fp, key))
                self._emit_local(self._record_of(fp, key), key)
        for o in range(self.world):
            if o != self.rank and self.sent[o]:
                v = [int(x) for x in verdicts[o].cpu()]
                assert len(v) == len(self.sent[o])
                for (fp, key), win in zip(self.sent[o], v):
                    if win:
                        self._emit_local(self._record_of(fp, key), key)     # the record stays with its generator
        return 0

    def count(self):
        return len(self.next_frontier), len(self.next_frontier)

    def empty_streams(self):
        z = torch.zeros(0, dtype=torch.int64)
        return (z, z, z)

    def export(self, first, n):
        words, off, fps = [], [], []
        for i in range(first, first + n):
            off.append(len(words))
            words.extend(int(w) for w in self.next_frontier[i])
            fps.append(self.next_fps[i])
        del self.next_frontier[first: first + n], self.next_fps[first: first + n]
        return (_i64(words), _i64(off), _i64(fps)), 0

    def _inv(self, s):
        if s["inv"]:
            self.viol_fp = min(self.viol_fp, s["fp"])
            self.viol_mask |= s["inv"]

    def _emit_local(self, s, key):
        self._inv(s)
        self.next_frontier.append(np.array(s["words"], dtype=np.uint64))
        self.next_fps.append(s["fp"])

    def append(self, words, off, fp):
        w, o, f = _u64(words), _u64(off), _u64(fp)
        bounds = o + [len(w)]
        for i in range(len(o)):
            self.next_frontier.append(np.array(w[bounds[i]: bounds[i + 1]], dtype=np.uint64))
            self.next_fps.append(f[i])
        return 0

    def commit(self):
        self.frontier, self.fps = self.next_frontier, self.next_fps
        self.level += 1
        self.total += len(self.frontier)
        return dict(n_new=len(self.frontier), expand_ms=0.0, materialize_ms=0.0, generated=self.generated, deadlocks=self.deadlocks,
                    pending=len(self.local_pending), viol_fp=self.viol_fp, viol_mask=self.viol_mask, max_bag=0)

    def probe(self):
        """the local part of the newest level, nothing stored: violating successors that are not in THIS rank's seen-set"""
        gen = dead = mask = 0
        bad = []
        for pidx, rec in enumerate(self.frontier):
            if rec is None:
                continue
            succ = orc.successors(self.P, rec)
            gen += len(succ)
            dead += 0 if succ else 1
            for k, s in enumerate(succ):
                if s["inv"] and s["fp"] not in self.seen:
                    bad.append((s["fp"], self._key(self.level + 1, s["auxkey"], self.fps[pidx], k)))
                    mask |= s["inv"]
        return dict(generated=gen, deadlocks=dead, viol_mask=mask), np.array(bad, dtype=np.uint64).reshape(-1, 2), 0

    def seen_before(self, fps, level):
        return np.array([int(f) in self.seen and (self.seen[int(f)] >> 55) < level for f in fps], dtype=bool)

    def loaded_level(self):
        return self.level

    def save(self, path):
        import pickle
        with open(path, "wb") as f:
            pickle.dump(dict(seen=self.seen, frontier=self.frontier, fps=self.fps, level=self.level, total=self.total), f)
        return 0

    @classmethod
    def load(cls, path, params, rank, world, owner_of):
        import pickle
        self = cls(params, rank, world, owner_of)
        with open(path, "rb") as f:
            d = pickle.load(f)
        self.seen, self.frontier, self.fps, self.level, self.total = d["seen"], d["frontier"], d["fps"], d["level"], d["total"]
        return self

    def lookup(self, key, level, by_low_bits):
        """one step of a trace walk through this rank's part of the seen-set -> (fingerprint, meta) or None"""
        if not by_low_bits:
            return (key, self.seen[key]) if key in self.seen else None
        hits = [(fp, meta) for fp, meta in self.seen.items() if (fp & ((1 << 45) - 1)) == key and (meta >> 55) == level]
        return (hits[0][0], hits[0][1], len(hits)) if hits else None

    def level_fps(self):
        return np.array(sorted(f for f in self.fps if f is not None), dtype=np.uint64)

    # ---- levels beyond the record buffers (the passes of csrc/vsr_deep.hpp, sharded: vsr_shard_loop.hpp's second half) ----------------
    # mode: "insert" (virtual level: claimed, counted, checked, nothing kept), "normal" (inserted and kept for the caller: a scratch buffer);
    # deep_regen (the rank rebuilds the states ITS candidates inserted — the winner set `won` — once per descent, asking nobody); deep_probe.
    TAKEN = 1

    def deep_source(self):
        """the newest stored level: [(record, fingerprint)]"""
        return [(r, f) for r, f in zip(self.frontier, self.fps) if r is not None]

    def deep_new_descent(self):
        """a new descent: every state of the winner set may be rebuilt once more (csrc: vsrmc_checker::wepoch)"""
        self.epoch = getattr(self, "epoch", 0) + 1

    def deep_regen(self, src, level):
        """src: [(record, fingerprint)] of level - 1 on this rank -> the states of `level` this rank's candidates inserted (its winner set:
        fingerprint -> [level, last descent]), each exactly once per descent, whichever of the rank's instances reaches it first.  The
        level matters: a successor of a level-l state may be ANOTHER level-l state of the set.  No exchange."""
        won = self.__dict__.setdefault("won", {})
        out = []
        for rec, _pfp in src:
            for s in orc.successors(self.P, rec):
                w = won.get(s["fp"])
                if w is not None and w[0] == level and w[1] != self.epoch:
                    w[1] = self.epoch
                    out.append((np.array(s["words"], dtype=np.uint64), s["fp"]))
        return out

    def _deep_claim_one(self, fp, key, level):
        """first inserter wins; keys of later candidates of the same level are min-merged -> True if this call inserted fp"""
        cur = self.seen.get(fp)
        if cur is None:
            self.seen[fp] = key
            return True
        if (cur >> 55) == level and key < cur:
            self.seen[fp] = key
        return False

    def deep_expand(self, src, level, mode):
        """src: [(record, fingerprint)] of level - 1 -> (per-owner (n, 2) candidate tensors, 0)"""
        if not hasattr(self, "sent_filter"):
            self.sent_filter = set()
        self.d_mode, self.d_level = mode, level
        self.d_out, self.d_sent, self.d_keep = [], [[] for _ in range(self.world)], [[] for _ in range(self.world)]
        st = self.d_stats = dict(generated=0, deadlocks=0, n_new=0, viol_fp=U64_MAX, viol_mask=0, fx=0, fs=0)
        for rec, pfp in src:
            succ = orc.successors(self.P, rec)
            st["generated"] += len(succ)
            st["deadlocks"] += 0 if succ else 1
            for k, s in enumerate(succ):
                key = self._key(level, s["auxkey"], pfp, k)
                o = self.owner_of(s["fp"], self.world)
                if o == self.rank:
                    if self._deep_claim_one(s["fp"], key, level):
                        self._deep_count(s)
                        if mode == "normal":
                            self.d_out.append((np.array(s["words"], dtype=np.uint64), s["fp"]))
                    continue
                tag = (s["fp"], s["auxkey"])                       # the sent-filter: an exact repeat was announced (and claimed) before
                if tag in self.sent_filter:
                    continue
                self.sent_filter.add(tag)
                self.d_sent[o].append((s["fp"], key))
                self.d_keep[o].append(s)                         # what the generator keeps beside the candidate
        return [_i64([x for fk in self.d_sent[o] for x in fk]).reshape(-1, 2) for o in range(self.world)], 0

    def _deep_count(self, s):
        self.__dict__.setdefault("won", {}).setdefault(s["fp"], [self.d_level, 0])   # this rank's candidate made the state: it regenerates it
        st = self.d_stats
        st["n_new"] += 1
        st["fx"] ^= s["fp"]
        st["fs"] = (st["fs"] + s["fp"]) & U64_MAX
        if s["inv"]:
            st["viol_fp"] = min(st["viol_fp"], s["fp"])
            st["viol_mask"] |= s["inv"]

    def deep_claim(self, cands, level, mode):
        vals = _u64(cands)
        out = []
        for i in range(len(vals) // 2):
            fp, key = vals[2 * i], vals[2 * i + 1]
            out.append(1 if self._deep_claim_one(fp, key, level) else 0)
        return torch.tensor(out, dtype=torch.uint8), 0

    def deep_apply(self, verdicts):
        """the verdict bytes of the announced candidates -> (records this pass yields [(record, fp)], the pass's figures)"""
        for o in range(self.world):
            if o == self.rank or not self.d_sent[o]:
                continue
            v = [int(x) for x in verdicts[o].cpu()]
            assert len(v) == len(self.d_sent[o])
            for s, win in zip(self.d_keep[o], v):
                if not win:
                    continue
                self._deep_count(s)
                if self.d_mode != "insert":
                    self.d_out.append((np.array(s["words"], dtype=np.uint64), s["fp"]))
        return self.d_out, self.d_stats

    def deep_probe(self, src, level):
        """successors of `src` (states of level - 1) that violate an invariant and are not in THIS rank's seen-set below `level`"""
        gen = dead = mask = 0
        bad = []
        for rec, pfp in src:
            succ = orc.successors(self.P, rec)
            gen += len(succ)
            dead += 0 if succ else 1
            for k, s in enumerate(succ):
                if s["inv"] and not (s["fp"] in self.seen and (self.seen[s["fp"]] >> 55) < level):
                    bad.append((s["fp"], self._key(level, s["auxkey"], pfp, k)))
                    mask |= s["inv"]
        return dict(generated=gen, deadlocks=dead, viol_mask=mask), bad

