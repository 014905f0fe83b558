"""oracle/tlcprint.py — prints the Python restatement's values in TLC's syntax (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

The inverse of oracle/tlcvalue.py, written without looking at the product's printer (csrc/vsr_format.hpp): what `tlc2.TLC -dump` writes
for a state — "State k:" and one "/\\ var = value" conjunct per variable — from the value shapes of oracle/pyoracle.py:
    tuple of (field, value) pairs -> [a |-> 1, b |-> x]      tuple -> <<a, b>>      frozenset -> {a, b}
    dict -> (k :> v @@ k2 :> v2), the empty one <<>> (as the reference's state_transfer_violation_trace.txt:8,11 prints it)
    bool -> TRUE / FALSE, int, str -> the bare identifier (every string of these models is a model value)
Element order inside sets / functions is this module's own (sorted by pyoracle.canon): TLC's order is its own business too, and a
reader of TLC values must not depend on it.  Used by tests/test_tlc_handoff.py to feed tools/diff_tlc_dump.py a dump that no part of
the product wrote, so that the tool is known to work before anyone with a JVM runs tools/tlc_handoff.sh."""
from oracle import pyoracle


def _is_record(x):
    return isinstance(x, tuple) and len(x) > 0 and all(isinstance(e, tuple) and len(e) == 2 and isinstance(e[0], str) for e in x) and \
        len({e[0] for e in x}) == len(x) and all(e[0].islower() or "_" in e[0] for e in x)


def fmt(x):
    if isinstance(x, bool):
        return "TRUE" if x else "FALSE"
    if isinstance(x, int):
        return str(x)
    if isinstance(x, str):
        return x
    if isinstance(x, (frozenset, set)):
        return "{" + ", ".join(fmt(e) for e in sorted(x, key=pyoracle.canon)) + "}"
    if isinstance(x, dict):
        if not x:
            return "<<>>"
        return "(" + " @@ ".join("%s :> %s" % (fmt(k), fmt(v)) for k, v in sorted(x.items(), key=lambda kv: pyoracle.canon(kv[0]))) + ")"
    if _is_record(x):
        return "[" + ", ".join("%s |-> %s" % (k, fmt(v)) for k, v in x) + "]"
    if isinstance(x, tuple):
        return "<<" + ", ".join(fmt(e) for e in x) + ">>"
    raise TypeError(type(x))


def dump_state(k, s, variables=None):
    """one state in the text form of `tlc2.TLC -dump`"""
    names = variables or (pyoracle.AUX_VARS + pyoracle.VIEW_VARS)
    return "State %d:\n" % k + "".join("/\\ %s = %s\n" % (v, fmt(s[v])) for v in sorted(names)) + "\n"


def dump(levels, variables=None):
    out, k = [], 0
    for lv in levels:
        for s in lv:
            k += 1
            out.append(dump_state(k, s, variables))
    return "".join(out)
