#!/bin/bash
# Branch census + instruction histograms of the secret-scalar kernels, from the gfx950 assembly hipcc emits
# (CPU only: compiles each translation unit to assembly once).  Output: profiles/<round>_isa_ct_<curve>_{base,gather,var}.txt
#   bash tools/isa_census_all.sh r03
ROUND=${1:-r03}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=/tmp/eccx_isa_$$; mkdir -p $TMP
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -mllvm -pragma-unroll-threshold=1000000 -S --cuda-device-only"
for u in k_p256 k_p384 k_p521 k_bls12_381 k_ed25519; do
  /opt/rocm/bin/hipcc $FLAGS $ROOT/eccoxide_amd/csrc/$u.hip -o $TMP/$u.s 2>/dev/null &
done
wait
emit() {  # unit curve-tag kernel-substring out-suffix
  { python3 $ROOT/tools/isa_histogram.py $1.hip "$3" --asm $TMP/$1.s --branches
    echo
    python3 $ROOT/tools/isa_histogram.py $1.hip "$3" --asm $TMP/$1.s --min-mads 300 | tail -n +3
  } > $ROOT/profiles/${ROUND}_isa_ct_$2_$4.txt
}
for spec in "k_p256 p256r1 P256U NoGlv" "k_p384 p384r1 P384U NoGlv" "k_p521 p521r1 P521U NoGlv" "k_bls12_381 bls12_381_g1 BLS12_381U NoGlv"; do
  set -- $spec
  emit $1 $2 "k_scalarmul_base_ct<eccx::$3, false>" base
  emit $1 $2 "k_scalarmul_base_ct<eccx::$3, true>" gather
  emit $1 $2 "k_scalarmul_coz_unsat<eccx::$3, eccx::$4, false, false, 4, true>" var
done
emit k_bls12_381 bls12_381_g1 "k_scalarmul_coz_unsat<eccx::BLS12_381U, eccx::BLS12_381_GLV, true, false, 4, true>" var_subgroup
emit k_ed25519 ed25519 "k_ed_scalarmul_base_ct<eccx::ED25519U, false>" base
emit k_ed25519 ed25519 "k_ed_scalarmul_base_ct<eccx::ED25519U, true>" gather
emit k_ed25519 ed25519 "k_ed_scalarmul_var_unsat<eccx::ED25519U, false, 3, true>" var
emit k_p256 p256r1 "k_batch_to_affine_unsat<eccx::P256U, 1, 16>" norm
rm -rf $TMP
ls $ROOT/profiles/${ROUND}_isa_ct_* | wc -l
