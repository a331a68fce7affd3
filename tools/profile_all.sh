#!/bin/bash
# Profile every bench workload on the GPU box (tools/profile.sh per workload) and summarise each
# into gpurun_out/<round>_<name>.json (copy those into profiles/).
# usage (inside gpurun): bash tools/profile_all.sh r02 [workload-spec ...]
#   a workload-spec is "name" or "name:variant"; default: every workload of bench.py
set -o pipefail
ROUND=${1:-r03}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
SPECS=("$@")
if [ ${#SPECS[@]} -eq 0 ]; then
  SPECS=("p256r1_var_2^20" "p384r1_var_2^19" "p521r1_var_2^19" "bls12_381_g1_var_2^20" "bls12_381_g1_var_2^20:glv" "ed25519_var_2^20" "p256r1_var_2^20:ct" "ed25519_var_2^20:ct" "p256r1_base_2^20:ct" "p256r1_base_2^20:ctg" "ed25519_base_2^20:ct" "ed25519_base_2^20:ctg"
         "p256r1_verify_2^20" "ed25519_base_2^20" "ed25519_base_2^20:lds" "p256r1_base_2^20" "x25519_2^20")
fi
for spec in "${SPECS[@]}"; do
  w=${spec%%:*}; v=default
  [[ "$spec" == *:* ]] && v=${spec##*:}
  tag=${ROUND}_$(echo "$w" | sed 's/\^//g')
  [ "$v" != default ] && tag=${tag}_$v
  echo "== $spec -> $tag"
  bash $REPO/tools/profile.sh $tag --workload "$w" --variant $v > $REPO/gpurun_out/prof_$tag.files 2>&1 || { echo "profile failed: $spec"; exit 1; }
  python3 $REPO/tools/prof_summary.py $REPO/gpurun_out/prof_$tag > $REPO/gpurun_out/$tag.json || exit 1
  # keep the merge-back small: the raw databases stay on the box
  rm -rf $REPO/gpurun_out/prof_$tag/trace $REPO/gpurun_out/prof_$tag/pmc_*/
done
ls -la $REPO/gpurun_out/${ROUND}_*.json
