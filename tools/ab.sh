#!/bin/bash
# A/B timing of two builds of libmi355_imgfilter.so on ONE box (boxes differ by +-10 %, so numbers from
# different gpurun calls are not comparable).  Usage: tools/ab.sh <libA.so> <rounds> -- <bench args> [-- <bench args> ...]
# <libA.so> may be a comma-separated list of builds (tools/build_variant.sh); each is labelled by its file name.
# Alternates the listed builds and B (= the in-tree build) `rounds` times per argument set; prints achieved GB/s.
A=$1; ROUNDS=$2; shift 2
[ "$1" = "--" ] && shift
sets=(); cur=""
for a in "$@"; do
  if [ "$a" = "--" ]; then sets+=("$cur"); cur=""; else cur="$cur $a"; fi
done
sets+=("$cur")
one() {  # $1 = label, $2 = lib or "", rest = args
  local label=$1 lib=$2; shift 2
  if [ -n "$lib" ]; then export MI355_IMGFILTER_LIB=$lib; else unset MI355_IMGFILTER_LIB; fi
  timeout -k 10 120 python bench.py --no-cpu-baseline --no-ceiling "$@" 2>/dev/null |
    python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', round(d['roofline']['achieved']), d['checksum'])" || echo "$label FAILED"
}
for s in "${sets[@]}"; do
  echo "== $s"
  for r in $(seq "$ROUNDS"); do
    for lib in ${A//,/ }; do one "$(basename $lib .so)" "$lib" $s; done
    one B "" $s
  done
done
