#!/bin/bash
# Profiles that belong to HEAD: refuses a dirty tree, stamps every summary with the commit it was taken from, and copies what is to be
# judged into profiles/ (gpurun_out/ is scratch).  Runs HERE (the build container): one gpurun call per configuration.
#   usage: scripts/run_profiles.sh TAG [c4 c2 c3 c5 c4_f32 c3_f32 c5_f32]        -> profiles/r06_TAG_{rocprof,bench}_<cfg>.{md,json}, profiles/pmc_<cfg>.json
set -eu
cd "$(dirname "$0")/.."
TAG=${1:?tag}; shift || true
CFGS=${*:-c4 c2 c3 c5}
if [ -n "$(git status --porcelain --untracked-files=no)" ]; then echo "the tree is dirty: commit first (a profile must name the commit it measures)"; git status --short | head; exit 2; fi
HEAD=$(git rev-parse --short=12 HEAD)
echo "$HEAD" > .profile_head
for cfg in $CFGS; do
  base=${cfg%_f32}; extra=""; [ "$cfg" != "$base" ] && extra="--dtype f32"
  case "$base" in
    c4) N=10000000; KERN=k_f_tridiag_store_wave;; c2) N=1000000; KERN=k_f_tridiag_fused;; c3) N=10000000; KERN=k_f_stencil5_store_wave;; c5) N=320000; KERN=k_f_blockcoupled_store;;
  esac
  [ "$cfg" = "c4_f32" ] && KERN=k_f_tridiag_store_wave4
  [ "$cfg" = "c3_f32" ] && KERN=k_f_stencil5_store_wave4
  [ "$cfg" = "c2_f32" ] && KERN=k_f_tridiag_fused4
  name=r06_${TAG}_$cfg
  /usr/local/graft/bin/gpurun --timeout 900 -- "bash scripts/profile_all.sh $name $base $N $KERN $extra" > gpurun_out/$name.log 2>&1 || true
  tail -2 gpurun_out/$name.log
  if [ -s gpurun_out/$name/summary.md ]; then
    { echo "head: $HEAD  (config $cfg; scripts/run_profiles.sh $TAG)"; echo; cat gpurun_out/$name/summary.md; } > profiles/${name/_$cfg/_rocprof_$cfg}.md
    [ -s gpurun_out/$name/bench.json ] && cp gpurun_out/$name/bench.json profiles/${name/_$cfg/_bench_$cfg}.json
    [ -s gpurun_out/$name/pmc.json ] && [ "$cfg" = "$base" ] && python - gpurun_out/$name/pmc.json profiles/pmc_$cfg.json "$HEAD" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); d["head"] = sys.argv[3]
json.dump(d, open(sys.argv[2], "w"), indent=1)
PY
  fi
done
rm -f .profile_head
ls -la profiles/ | grep "r06_${TAG}" || true
