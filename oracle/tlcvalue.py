"""oracle/tlcvalue.py — parser for TLC's printed value syntax (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

Reads the "trace expression" dump format of /root/reference/state_transfer_violation_trace.txt (a TLA+
sequence `<< [ _TEAction |-> [...], var |-> value, ... ], ... >>`) into the Python value shapes used by
oracle/pyoracle.py:
    record   [a |-> 1, b |-> x]     -> tuple(sorted((field, value)))
    sequence <<a, b>>               -> tuple
    set      {a, b}                 -> frozenset
    function (k :> v @@ k2 :> v2)   -> dict            (also `k :> v` alone inside parentheses)
    interval 1..3                   -> frozenset({1, 2, 3})
    TRUE / FALSE / 12 / "str" / Ident
"""
import re

_TOKEN = re.compile(r"""\s*(?:
      (?P<num>\d+)
    | (?P<str>"[^"]*")
    | (?P<id>[A-Za-z_][A-Za-z_0-9]*)
    | (?P<op><<|>>|\|->|:>|@@|\.\.|[\[\]\{\}\(\),])
    )""", re.X)


def tokenize(text):
    pos, out = 0, []
    n = len(text)
    while pos < n:
        m = _TOKEN.match(text, pos)
        if not m:
            if text[pos:].strip() == "":
                break
            raise ValueError("cannot tokenize at %d: %r" % (pos, text[pos:pos + 40]))
        pos = m.end()
        if m.group("num") is not None:
            out.append(("num", int(m.group("num"))))
        elif m.group("str") is not None:
            out.append(("str", m.group("str")[1:-1]))
        elif m.group("id") is not None:
            out.append(("id", m.group("id")))
        else:
            out.append(("op", m.group("op")))
    return out


class _P:
    def __init__(self, toks):
        self.t, self.i = toks, 0

    def peek(self):
        return self.t[self.i] if self.i < len(self.t) else (None, None)

    def eat(self, kind=None, val=None):
        k, v = self.peek()
        if (kind is not None and k != kind) or (val is not None and v != val):
            raise ValueError("expected %r %r, got %r %r at token %d" % (kind, val, k, v, self.i))
        self.i += 1
        return v

    def value(self):
        v = self.atom()
        k, o = self.peek()
        if k == "op" and o == "..":                      # interval
            self.eat()
            hi = self.atom()
            return frozenset(range(v, hi + 1))
        if k == "op" and o == ":>":                      # function literal k :> v @@ ...
            self.eat()
            d = {v: self.value_no_fn()}
            while self.peek() == ("op", "@@"):
                self.eat()
                key = self.atom()
                self.eat("op", ":>")
                d[key] = self.value_no_fn()
            return d
        return v

    def value_no_fn(self):
        v = self.atom()
        if self.peek() == ("op", ".."):
            self.eat()
            return frozenset(range(v, self.atom() + 1))
        return v

    def atom(self):
        k, v = self.peek()
        if k == "num" or k == "str":
            self.eat()
            return v
        if k == "id":
            self.eat()
            if v == "TRUE":
                return True
            if v == "FALSE":
                return False
            return v
        if k == "op" and v == "<<":
            self.eat()
            items = []
            while self.peek() != ("op", ">>"):
                items.append(self.value())
                if self.peek() == ("op", ","):
                    self.eat()
            self.eat("op", ">>")
            return tuple(items)
        if k == "op" and v == "{":
            self.eat()
            items = []
            while self.peek() != ("op", "}"):
                items.append(self.value())
                if self.peek() == ("op", ","):
                    self.eat()
            self.eat("op", "}")
            return frozenset(items)
        if k == "op" and v == "(":
            self.eat()
            val = self.value()
            self.eat("op", ")")
            return val
        if k == "op" and v == "[":
            self.eat()
            fields = []
            while self.peek() != ("op", "]"):
                name = self.eat("id")
                self.eat("op", "|->")
                fields.append((name, self.value()))
                if self.peek() == ("op", ","):
                    self.eat()
            self.eat("op", "]")
            return tuple(sorted(fields))
        raise ValueError("unexpected token %r %r at %d" % (k, v, self.i))


def parse_value(text):
    p = _P(tokenize(text))
    v = p.value()
    if p.i != len(p.t):
        raise ValueError("trailing tokens")
    return v


def parse_trace(text):
    """-> list of (action name, {var: value}) for each state of a TLC trace-expression dump."""
    seq = parse_value(text)
    out = []
    for st in seq:
        d = dict(st)
        act = dict(d.pop("_TEAction"))
        # `<<>>` prints both the empty sequence and the empty function; normalise the two function-valued vars
        for fv in ("messages", "aux_client_acked"):
            if d.get(fv) == ():
                d[fv] = {}
        out.append((act["name"], act["position"], d))
    return out
