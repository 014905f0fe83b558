#!/bin/bash
# tools/tlc_handoff.sh — pin this checker against TLC itself, on a box that has BOTH a JVM with tla2tools.jar and an MI355X.
#
#   tools/tlc_handoff.sh /path/to/tla2tools.jar /path/to/vsr-tlaplus   (the reference checkout: vsr-revisited/paper/VSR.tla)
#
# The build image has no JVM and the reference pins no TLC version, counts or fingerprints ("parity against TLC itself is
# unpinned").  This script is the whole hand-off: for BASELINE configs[0] and [1] (and the two analysis models) it runs
#   java -cp tla2tools.jar tlc2.TLC -workers 1 -deadlock -fp 0 -dump states.dump -config X.cfg X.tla
# and holds three things against it:
#   1. the distinct-state count TLC prints against `vsrmc` on the same cfg,
#   2. the SET of states: tools/diff_tlc_dump.py reads the dump with the product's TLC reader, fingerprints every dumped state under
#      VIEW + SYMMETRY on the GPU and compares with the GPU BFS (independent of any recalled TLC constant),
#   3. (VSR.tla only) TLC's own FP64 values: the dump's states through vsrmc_tlc_fingerprint_batch against the fingerprints TLC prints
#      with -dump dot / the trace explorer, when the TLC build at hand can print them — the mode whose constants are [TLC-RECALLED].
# tests/test_tlc_handoff.py runs step 2 against a dump written by the Python restatement in TLC's syntax (oracle/tlcprint.py), so the
# differ is known to work before anyone with a JVM tries this.
set -e
JAR=${1:?usage: tools/tlc_handoff.sh tla2tools.jar reference-checkout}
REF=${2:?usage: tools/tlc_handoff.sh tla2tools.jar reference-checkout}
HERE=$(cd "$(dirname "$0")/.." && pwd)
OUT=${OUT:-$HERE/gpurun_out/tlc_handoff}
mkdir -p "$OUT"
command -v java >/dev/null || { echo "no java on PATH"; exit 2; }
run_one() {   # label, tla, cfg text
  local label=$1 tla=$2 cfg=$OUT/$1.cfg
  printf '%s\n' "$3" > "$cfg"
  cp "$tla" "$OUT/"; cp "$(dirname "$tla")"/*.tla "$OUT/" 2>/dev/null || true
  echo "== $label: TLC"
  (cd "$OUT" && java -XX:+UseParallelGC -cp "$JAR" tlc2.TLC -workers 1 -deadlock -fp 0 -dump "$label.dump" -config "$label.cfg" "$(basename "$tla")" > "$label.tlc.log" 2>&1) || true
  grep -E "distinct states found|is violated|Error" "$OUT/$label.tlc.log" | head -5
  echo "== $label: vsrmc"
  "$HERE/vsr_tlaplus_amd/vsrmc" -config "$cfg" "$tla" > "$OUT/$label.vsrmc.log" 2>&1 || true
  grep -E "distinct states found|is violated|Error" "$OUT/$label.vsrmc.log" | head -5
  local a b
  a=$(grep -oE "[0-9]+ distinct states found" "$OUT/$label.tlc.log" | tail -1 | grep -oE "^[0-9]+")
  b=$(grep -oE "[0-9]+ distinct states found" "$OUT/$label.vsrmc.log" | tail -1 | grep -oE "^[0-9]+")
  echo "$label: distinct states  TLC=$a  vsrmc=$b  $([ "$a" = "$b" ] && echo EQUAL || echo DIFFERENT)"
  echo "== $label: state sets"
  python "$HERE/tools/diff_tlc_dump.py" -config "$cfg" -tla "$tla" "$OUT/$label.dump" --subset-ok | tail -4
}
VSR=$REF/vsr-revisited/paper/VSR.tla
cfg_vsr() { cat <<CFG
CONSTANTS
    ReplicaCount = $1
    ClientCount = 1
    Values = {$2}
    StartViewOnTimerLimit = $3
    RestartEmptyLimit = 0
    Normal = Normal
    ViewChange = ViewChange
    Recovering = Recovering
    v1 = v1
    v2 = v2
    v3 = v3
    Nil = Nil
    AnyDest = AnyDest
    PrepareMsg = PrepareMsg
    PrepareOkMsg = PrepareOkMsg
    StartViewChangeMsg = StartViewChangeMsg
    DoViewChangeMsg = DoViewChangeMsg
    StartViewMsg = StartViewMsg
    GetStateMsg = GetStateMsg
    NewStateMsg = NewStateMsg
    RecoveryMsg = RecoveryMsg
    RecoveryResponseMsg = RecoveryResponseMsg
INIT Init
NEXT Next
VIEW view
SYMMETRY symmValues
INVARIANT AcknowledgedWriteNotLost
CFG
}
# BASELINE configs[0]: (2,1,{v1},1) — 76 states; a (2,1,{v1,v2},2) run in between — 2 073 states, symmetry on; configs[1] = the shipped VSR.cfg
run_one config1 "$VSR" "$(cfg_vsr 2 v1 1)"
run_one small22 "$VSR" "$(cfg_vsr 2 'v1, v2' 2)"
echo "== shipped VSR.cfg (319 M states to the violation at depth 28: TLC needs hours and > 100 GB; run it if the box has them)"
echo "   java -cp $JAR tlc2.TLC -workers auto -deadlock -config $REF/vsr-revisited/paper/VSR.cfg $VSR"
echo "   expected from this checker: 319228361 distinct states, AcknowledgedWriteNotLost violated at depth 28 (bench.py's config2 object)"
echo "logs under $OUT"
